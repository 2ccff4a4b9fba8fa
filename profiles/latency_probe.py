"""p50 / p99 wall time of one search_batch call of 1, 8 and 32 queries on the bench corpus (launch + device sync included)."""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ravqa_amd
from ravqa_amd import synth
from ravqa_amd.scorer import IndexScorer
corpus = synth.make_corpus(1_000_000, 128, 131072, 2, seed=0, device="cuda")
Q, _ = synth.make_queries(corpus, 64, 32, seed=2)
sc = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=256)
for bsz in (1, 8, 32):
    ts = []
    for i in range(220):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sc.search_batch(Q[i % 32:i % 32 + bsz], 100, 2, 0.45, 1024, 32)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts = sorted(ts[20:]); print(bsz, round(statistics.median(ts), 4), round(ts[int(0.99 * (len(ts) - 1))], 4))
sc.check()
