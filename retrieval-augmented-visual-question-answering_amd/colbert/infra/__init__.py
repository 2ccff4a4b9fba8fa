from ravqa_amd.config import ColBERTConfig, Run, RunConfig  # noqa: F401  (TPC/infra/__init__.py)
