"""Workload statistics of the candidate / stage-1 scatter kernel at the bench shape: qualifying centroids per query,
(centroid, passage) pairs walked, candidates that are in the hit set, per 32768-passage chunk."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from ravqa_amd import synth, _native
from ravqa_amd.scorer import IndexScorer
corpus = synth.make_corpus(1_000_000, 128, 131072, 2, seed=0, device="cuda")
Q, _ = synth.make_queries(corpus, 64, 32, seed=2)
sc = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=64)
sc.search_batch(Q, 100, 2, 0.45, 1024, 32)
torch.cuda.synchronize()
ivf_len = corpus.ivf_lengths.cpu().numpy().astype(np.int64)
off = np.concatenate([[0], np.cumsum(ivf_len)])
ivf = corpus.ivf.cpu().numpy()
rows = []
for q in range(0, 64, 8):
    bits = sc.tap(_native.TAP_IDX_BITS, q)
    qual = np.nonzero(np.unpackbits(bits.view(np.uint8), bitorder="little"))[0]
    cells = sc.tap(_native.TAP_CELLS, q)
    cand = sc.tap(_native.TAP_CANDIDATES, q)
    pairs_c = int(ivf_len[cells].sum()); pairs_q = int(ivf_len[qual].sum())
    hit = np.unique(np.concatenate([ivf[off[c]:off[c + 1]] for c in qual])) if len(qual) else np.zeros(0, np.int32)
    nh = np.intersect1d(hit, cand).size
    per_chunk = np.bincount(np.intersect1d(hit, cand) // 32768, minlength=31)
    rows.append((len(cells), len(qual), pairs_c, pairs_q, len(cand), nh, int(per_chunk.max())))
    print(rows[-1])
print("mean: cells, qual, cell pairs, qual pairs, candidates, cand&hit, max cand&hit per chunk\n", np.mean(rows, axis=0))
