"""Drop-in for the reference's ivector_PLDA_OSI.py (same module and class name)."""
from fakebob_amd.systems import iv_OSI  # noqa: F401
