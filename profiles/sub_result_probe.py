"""One of bench.py's sub-result workloads alone, a few steps -- the command rocprofv3 wraps for the per-sub-result kernel statistics
under profiles/rNN/ (profiles/measure_subresults.sh):
    python profiles/sub_result_probe.py k500 | thr0.25 | nq320 | nq832 | built256 | built4096 | headline
Prints one JSON line (queries/s, ms per step, per-stage HIP-event times)."""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import ravqa_amd  # noqa: F401
from ravqa_amd import indexing, synth
from ravqa_amd.scorer import IndexScorer

what = sys.argv[1] if len(sys.argv) > 1 else "headline"
nq = {"nq320": 320, "nq832": 832}.get(what, 32)
k, pol = {"k500": (500, (4, 0.4, 4096)), "thr0.25": (100, (2, 0.25, 1024))}.get(what, (100, (2, 0.45, 1024)))
if what.startswith("built"):
    embs, doclens, planted = synth.make_overlapping_embeddings(1_000_000, 128, int(what[5:]), seed=0, device="cuda")
    arrays = indexing.build_index(embs, doclens, nbits=2, kmeans_niters=4)
    del embs
    scorer = IndexScorer(arrays=arrays, max_batch=256)
    Qs = [planted(1024)[0] for _ in range(2)]
else:
    corpus = synth.make_corpus(1_000_000, 128, 131072, 2, seed=0, device="cuda")
    scorer = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=256)
    Qs = [synth.make_queries(corpus, 1024, nq, seed=40 + j)[0] for j in range(2)]
for i in range(2):
    scorer.search_batch(Qs[i % 2], k, pol[0], pol[1], pol[2], 32)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 4
for i in range(n):
    scorer.search_batch(Qs[i % 2], k, pol[0], pol[1], pol[2], 32)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
scorer.check()
for i in range(2):
    scorer.search_batch(Qs[i % 2], k, pol[0], pol[1], pol[2], 32, profile=True)
st = {a: round(b / 2, 3) for a, b in scorer.stage_ms().items()}
print(json.dumps({"workload": what, "policy": list(pol) + [k], "nq": nq, "queries_per_sec": round(1024 / dt), "ms_per_step": round(dt * 1e3, 3), "stage_ms": st}))
