from ravqa_amd.data import Collection  # noqa: F401
