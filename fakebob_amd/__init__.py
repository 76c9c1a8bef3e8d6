"""fakebob_amd -- MI355X-native engine for the FAKEBOB NES attack hot path.

Only what the path needs: csrc/ (HIP kernels + the C ABI), the ctypes binding
and the host-side mirror of the reference's FakeBob / model-wrapper interface.
"""
__version__ = "0.1.0"
