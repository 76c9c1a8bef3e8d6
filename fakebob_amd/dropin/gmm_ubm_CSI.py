"""Drop-in for the reference's gmm_ubm_CSI.py (same module and class name)."""
from fakebob_amd.systems import gmm_CSI  # noqa: F401
