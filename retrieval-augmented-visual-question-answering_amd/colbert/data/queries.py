from ravqa_amd.data import Queries  # noqa: F401
