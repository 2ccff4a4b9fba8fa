from ravqa_amd.scoring import ColBERT, colbert_score, colbert_score_packed, colbert_score_reduce  # noqa: F401
