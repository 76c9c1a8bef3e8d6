"""Drop-in for the reference's gmm_ubm_OSI.py (same module and class name)."""
from fakebob_amd.systems import gmm_OSI  # noqa: F401
