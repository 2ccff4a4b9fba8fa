from ravqa_amd.data import Ranking  # noqa: F401
