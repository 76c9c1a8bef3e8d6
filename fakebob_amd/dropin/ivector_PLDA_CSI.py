"""Drop-in for the reference's ivector_PLDA_CSI.py (same module and class name)."""
from fakebob_amd.systems import iv_CSI  # noqa: F401
