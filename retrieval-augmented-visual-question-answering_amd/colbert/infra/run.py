from ravqa_amd.config import Run  # noqa: F401
