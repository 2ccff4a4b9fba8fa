from ravqa_amd.config import ColBERTConfig, RunConfig  # noqa: F401
