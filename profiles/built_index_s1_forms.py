"""Stage-1 forms on the 1 M-passage index built from 256 overlapping topics (~1.5 k surviving centroids per query):
FLMR_S1_IMPL = narrow (one fp32 row per half-wave and load) | unset (wide: four hits per passage and load) | rank (rank rows in
LDS) | rankg (rank rows through L2)."""
import sys, json
sys.path.insert(0, 'profiles'); sys.path.insert(0, '.')
import torch, ravqa_amd
from ravqa_amd import indexing, synth, _native
from ravqa_amd.scorer import IndexScorer
embs, doclens, planted = synth.make_overlapping_embeddings(1_000_000, 128, 256, seed=0, device="cuda")
arrays = indexing.build_index(embs, doclens, nbits=2, kmeans_niters=4)
del embs
Q, tgt = planted(1024)
import os
# (the rank / wide / ablation forms lived on a scratch branch of csrc/flmr_filter.hip: profiles/r05/built_index_s1_forms.txt; with the
# tree's library every FLMR_S1_IMPL value below runs the same code-scanning kernel)
for impl in (os.environ.get("S1_FORMS", "").split(",") if os.environ.get("S1_FORMS") else (None,)):
    if impl: _native.set_option("FLMR_S1_IMPL", impl)
    else: _native.set_option("FLMR_S1_IMPL", None)
    sc = IndexScorer(arrays=arrays, max_batch=256)
    for _ in range(2): sc.search_batch(Q, 100, 2, 0.45, 1024, 32)
    sc.search_batch(Q, 100, 2, 0.45, 1024, 32, profile=True)
    st = sc.stage_ms()
    print(impl, {k: round(v, 2) for k, v in st.items() if v > 0.5})
    sc.close_searcher()
