"""Drop-in for the reference's FAKEBOB.py (same module and class name).  The attack itself has no pipeline options: the
round trips of the reference's scoring pipeline belong to the system classes (gmm_ubm_OSI.py ... in this directory)."""
from fakebob_amd.attack import FakeBob, UNTARGETED  # noqa: F401
