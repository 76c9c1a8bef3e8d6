"""Drop-in for the reference's ivector_PLDA_SV.py (same module and class name)."""
from fakebob_amd.systems import iv_SV  # noqa: F401
from fakebob_amd.systems import use_reference_pipeline_defaults as _ref_defaults

_ref_defaults()  # this module name is the reference's: behave like its pipeline (dropin/README.md)
