"""Per-rank compute time of the exact sharded protocol for world sizes 1..8, measured on ONE GPU.

Rank 0's share of the work (its 1/W passage shard, its 1/W slice of the queries for the query-split stage 0) is executed
for real; the all-gathers are replaced by repeating rank 0's own buffers W times with the pids shifted into the other ranks' ranges (right sizes and
survivor shares, made-up contents for the other ranks' parts -- timing only, results are not checked here; tests/test_hip_parity.py does that).  What is missing from the
figures is only the RCCL time of 4 small all-gathers per step.  Usage: python profiles/shard_step_model.py [passages [worlds, e.g. 1,2,4,8 [queries per step]]]
"""
import sys, time, json
import torch
sys.path.insert(0, ".")
import ravqa_amd as pkg  # noqa: F401
from ravqa_amd import synth, ops
from ravqa_amd.scorer import IndexScorer

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
DOCLEN, NB, NQ, k = 128, 2, 32, 100
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256   # queries per step
ncells, thr, ndocs = 2, 0.45, 1024
K = 2 ** int(torch.log2(torch.tensor(16.0 * ((P * DOCLEN) ** 0.5))).floor())
corpus = synth.make_corpus(P, DOCLEN, K, NB, seed=0, device="cuda")
Q, _ = synth.make_queries(corpus, B, NQ, seed=2)


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


out = {}
WORLDS = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 2, 4, 8)
for W in WORLDS:
    sh = corpus if W == 1 else synth.shard_corpus(corpus, 0, W)
    sc = IndexScorer(device_index=synth.corpus_device_index(sh, pid_base=0), max_batch=B)
    per = -(-B // W)

    def exchange(keys, n, ordered=False):
        # the other ranks' rows: the same keys with their pids moved into that rank's pid range (same scores, disjoint pids),
        # so that about 1/W of each global survivor set belongs to this shard -- as in a real run; each shard ships its
        # phase1_width(ndocs, W) best keys like distributed.py's default exchange
        from ravqa_amd.distributed import phase1_width
        m = phase1_width(n, W)
        if m < keys.size(1):
            keys = ops.topn_keys(keys, m, ordered=False)
        shift = (torch.arange(W, device="cuda", dtype=torch.int64) * (P // W)).view(W, 1, 1)
        g = torch.where(keys.unsqueeze(0) != 0, keys.unsqueeze(0) + shift, torch.zeros_like(keys).unsqueeze(0))
        return ops.topn_keys(g.permute(1, 0, 2).reshape(g.size(1), -1), n, ordered=ordered)

    def step(split):
        if split and W > 1:
            bits, cells, ncell = sc.probe(Q, k, ncells, thr, ndocs, 0, per, 32)
            bits, cells, ncell = (t.repeat((W,) + (1,) * (t.dim() - 1))[:B] for t in (bits, cells, ncell))
            k1 = sc.phase1_probed(Q, k, ncells, thr, ndocs, bits, cells, ncell, 32)
        else:
            k1 = sc.phase1(Q, k, ncells, thr, ndocs, 32)
        s1 = exchange(k1, ndocs)
        s2 = ops.topn_keys(sc.phase2(s1), ndocs // 4, ordered=False)      # (a SUM all-reduce leaves the array size unchanged)
        fin = ops.topn_keys(sc.phase3(s2), k, ordered=True)
        return ops.unpack_keys(fin, k)

    rec = {"replicated_stage0_ms": timed(lambda: step(False))}
    if W > 1:
        rec["query_split_stage0_ms"] = timed(lambda: step(True))
    if W == 1:
        rec["unsharded_search_batch_ms"] = timed(lambda: sc.search_batch(Q, k, ncells, thr, ndocs, 32))
    out[W] = rec
    print(W, rec, flush=True)
    del sc
print(json.dumps(out))
