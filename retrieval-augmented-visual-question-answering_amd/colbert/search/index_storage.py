from ravqa_amd.scorer import IndexScorer  # noqa: F401  (TPC/search/index_storage.py)
