"""Drop-in for the reference's FAKEBOB.py (same module and class name)."""
from fakebob_amd.attack import FakeBob, UNTARGETED  # noqa: F401
