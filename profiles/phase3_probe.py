"""Time phase 3 (exact MaxSim of a shard's members) as a function of the number of members per query, on one GPU:
how the fused MaxSim kernel behaves when a query has few documents (the per-rank situation at N shards: 256 / N)."""
import sys, time
import torch
sys.path.insert(0, ".")
import ravqa_amd  # noqa: F401
from ravqa_amd import synth
from ravqa_amd.scorer import IndexScorer

P, B = 1_000_000, 1024
K = 131072
corpus = synth.make_corpus(P, 128, K, 2, seed=0, device="cuda")
Q, _ = synth.make_queries(corpus, B, 32, seed=2)
scorer = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=B)
keys1 = scorer.phase1(Q, 100, 2, 0.45, 1024, 32)          # sets the phase state
g = torch.Generator(device="cuda").manual_seed(1)
for members in (256, 128, 64, 32, 16):
    pids = torch.randint(0, P, (B, 256), generator=g, device="cuda", dtype=torch.int64)
    keys = (torch.full((B, 256), 0x3F800000, dtype=torch.int64, device="cuda") << 32) | pids
    keys[:, members:] = 0                                   # slots of other shards
    for _ in range(2):
        scorer.phase3(keys)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        scorer.phase3(keys)
    torch.cuda.synchronize()
    print(f"members per query {members:4d}: phase 3 {1e3 * (time.perf_counter() - t0) / 10:.3f} ms per {B} queries")
