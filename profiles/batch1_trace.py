"""Kernel trace of ONE-query calls (the reference's calling pattern, searcher.py:73-89) on the bench corpus:
   cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o s -- python $REPO/profiles/batch1_trace.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ravqa_amd
from ravqa_amd import synth
from ravqa_amd.scorer import IndexScorer

corpus = synth.make_corpus(1_000_000, 128, 131072, 2, seed=0, device="cuda")
scorer = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=32)
Q, _ = synth.make_queries(corpus, 64, 32, seed=3)
for i in range(8):
    scorer.search_batch(Q[i:i + 1], 100, 2, 0.45, 1024, 32)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 200
for i in range(N):
    scorer.search_batch(Q[i % 64:i % 64 + 1], 100, 2, 0.45, 1024, 32)
    torch.cuda.synchronize()
print("ms per call", (time.perf_counter() - t0) / N * 1e3)
