from ravqa_amd.data import Provenance  # noqa: F401
