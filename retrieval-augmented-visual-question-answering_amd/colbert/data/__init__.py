from ravqa_amd.data import Collection, Queries, Ranking  # noqa: F401  (TPC/data/__init__.py)
