"""How many surviving centroids does a hit candidate contain?  (sizing of the per-passage accumulators of the scatter stage 1)"""
import sys, torch, numpy as np
sys.path.insert(0, ".")
import ravqa_amd
from ravqa_amd import synth, _native
from ravqa_amd.scorer import IndexScorer
P, K, B = 1_000_000, 131072, 8
corpus = synth.make_corpus(P, 128, K, 2, seed=0, device="cuda")
Q, _ = synth.make_queries(corpus, B, 32, seed=2)
sc = IndexScorer(device_index=synth.corpus_device_index(corpus, pid_base=0), max_batch=B)
sc.search_batch(Q, 100, 2, 0.45, 1024, 32)
off = corpus.ivf_offsets.cpu().numpy()
ivf = corpus.ivf.cpu().numpy()
for q in range(B):
    bits = sc.tap(_native.TAP_IDX_BITS, q)
    qual = np.nonzero(np.unpackbits(bits.view(np.uint8), bitorder="little"))[0]
    cand = sc.tap(_native.TAP_CANDIDATES, q)
    pairs = np.concatenate([ivf[off[c]:off[c + 1]] for c in qual])
    pairs = pairs[np.isin(pairs, cand)]
    u, cnt = np.unique(pairs, return_counts=True)
    print(f"query {q}: surviving centroids {qual.size}, candidates {cand.size}, hit candidates {u.size}, (centroid, passage) pairs {pairs.size}, "
          f"with 1 / 2 / 3+ surviving centroids: {np.mean(cnt == 1):.3f} / {np.mean(cnt == 2):.3f} / {np.mean(cnt >= 3):.3f}")
