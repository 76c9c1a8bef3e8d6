"""Drop-in for the reference's ivector_PLDA_OSI.py: the same class name, with the reference pipeline's two file round trips
(CompressedMatrix MFCC storage, 6-digit score text) on by default -- a subclass, nothing global is switched
(dropin/README.md)."""
from fakebob_amd import systems as _systems

iv_OSI = _systems.reference_pipeline(_systems.iv_OSI, __name__)
