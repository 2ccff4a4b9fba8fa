from ravqa_amd.searcher import Searcher  # noqa: F401  (TPC/searcher.py)
