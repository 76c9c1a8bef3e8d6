"""Drop-in for the reference's gmm_ubm_SV.py (same module and class name)."""
from fakebob_amd.systems import gmm_SV  # noqa: F401
