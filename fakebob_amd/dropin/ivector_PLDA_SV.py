"""Drop-in for the reference's ivector_PLDA_SV.py (same module and class name)."""
from fakebob_amd.systems import iv_SV  # noqa: F401
