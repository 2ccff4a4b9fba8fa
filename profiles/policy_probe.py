"""Stage split of one 1024-query step at several pruning policies on the bench corpus (1 M x 128, K = 131072, nbits 2):
    python profiles/policy_probe.py [ncells,thr,ndocs,k ...]
Default: the k <= 100 policy, two lower thresholds (thousands of surviving centroids per query: the code-scanning stage 1) and
the k = 500 policy.  Prints queries/s, ms per step and the per-stage HIP-event times; FLMR_HIP_LIB selects a library variant."""
import os, sys, time, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import ravqa_amd  # noqa: F401
from ravqa_amd import synth
from ravqa_amd.scorer import IndexScorer

pols = [tuple(float(x) if "." in x else int(x) for x in a.split(",")) for a in sys.argv[1:]] or \
       [(2, 0.45, 1024, 100), (2, 0.35, 1024, 100), (2, 0.25, 1024, 100), (4, 0.4, 4096, 500)]
npass = int(os.environ.get("PROBE_PASSAGES", 1_000_000))
K = int(os.environ.get("PROBE_K", 131072))
corpus = synth.make_corpus(npass, 128, K, 2, seed=0, device="cuda")
scorer = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=256)
Qs = [synth.make_queries(corpus, 1024, 32, seed=2 + j)[0] for j in range(2)]
for (ncells, thr, ndocs, k) in pols:
    for i in range(2):
        scorer.search_batch(Qs[i % 2], k, ncells, thr, ndocs, 32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 4
    for i in range(n):
        scorer.search_batch(Qs[i % 2], k, ncells, thr, ndocs, 32)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    scorer.check()
    for i in range(2):
        scorer.search_batch(Qs[i % 2], k, ncells, thr, ndocs, 32, profile=True)
    st = {a: round(b / 2, 3) for a, b in scorer.stage_ms().items()}
    nsurv = sum(bin(int(x)).count("1") for x in scorer.tap(ravqa_amd._native.TAP_IDX_BITS, 0))
    print(json.dumps({"policy": [ncells, thr, ndocs, k], "queries_per_sec": round(1024 / dt), "ms_per_step": round(dt * 1e3, 3),
                      "surviving_centroids_q0": nsurv, "stage_ms": st}))
