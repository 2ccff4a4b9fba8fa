"""N > 1 path on CPU: two gloo ranks, each owning a passage shard (IndexArrays.shard), exchange their per-shard top-k
with ShardedSearcher's all-gather and merge.  The per-shard search is INJECTED (the CPU oracle on the shard's arrays) and
so is the merge (a torch reference of flmr_merge_topk): what is under test here is the host logic of the product --
shard arithmetic (local/global pids, IVF restriction), the exchange layout, the merge contract -- not the kernels, which
only run on the MI355X (tests/test_hip_parity.py::test_merge_topk covers the HIP merge)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def merge_topk_reference(scores, pids):
    """[R, n, k] -> top-k per query by (score, pid) descending; -1 pids are empty slots."""
    R, n, k = scores.shape
    out_s = torch.zeros((n, k), dtype=torch.float32)
    out_p = torch.full((n, k), -1, dtype=torch.int32)
    out_c = torch.zeros(n, dtype=torch.int32)
    for q in range(n):
        items = [(float(scores[r, q, i]), int(pids[r, q, i])) for r in range(R) for i in range(k) if int(pids[r, q, i]) >= 0]
        items.sort(reverse=True)
        for i, (s, p) in enumerate(items[:k]):
            out_s[q, i], out_p[q, i] = s, p
        out_c[q] = min(k, len(items))
    return out_s, out_p, out_c


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ravqa_amd
    from ravqa_amd.distributed import ShardedSearcher
    from oracle import oracle as orc
    z = dict(np.load(os.path.join(ROOT, "tests", "golden", "idx_nb2.npz")))
    full = ravqa_amd.IndexArrays.from_golden(z)
    shard = full.shard(rank, world)
    oi = orc.OracleIndex(shard.dim, shard.nbits, shard.codes, shard.residuals, shard.doclens, shard.ivf, shard.ivf_lengths,
                         shard.centroids, shard.bucket_weights)
    k, ncells, thr, ndocs = 10, 2, 0.45, 256

    def local_search(Q, k, **kw):
        p, s, c = oi.search_batch(Q.numpy(), k, ncells, thr, ndocs)
        p = np.where(p >= 0, p + shard.pid_base, -1)  # global pids, as flmr_search_batch returns them
        return torch.from_numpy(p.astype(np.int32)), torch.from_numpy(s), torch.from_numpy(c)

    ss = ShardedSearcher(local_search=local_search, merge=merge_topk_reference)
    assert (ss.rank, ss.world) == (rank, world)
    Q = torch.stack([torch.from_numpy(z[f"rank{i}.Q"]) for i in (0, 3)])
    pids, scores, counts = ss.search_batch(Q, k)
    lp, ls, lc = local_search(Q, k)
    torch.save({"pids": pids, "scores": scores, "counts": counts, "local_pids": lp, "local_scores": ls,
                "pid_base": shard.pid_base, "n_local": shard.num_passages}, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_search(tmp_path):
    world, port = 2, _free_port()
    out = str(tmp_path / "res")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    res = [torch.load(f"{out}.{r}") for r in range(world)]
    # every rank ends with the same global ranking
    assert torch.equal(res[0]["pids"], res[1]["pids"]) and torch.equal(res[0]["scores"], res[1]["scores"])
    # shards tile the pid space and return global ids inside their own range
    assert res[0]["pid_base"] == 0 and res[1]["pid_base"] == res[0]["n_local"]
    for r in res:
        lp = r["local_pids"]
        ok = lp[lp >= 0]
        assert int(ok.min()) >= r["pid_base"] and int(ok.max()) < r["pid_base"] + r["n_local"]
    # merged list == top-k of the union of the per-shard lists ("fast mode")
    gs = torch.stack([r["local_scores"] for r in res])
    gp = torch.stack([r["local_pids"] for r in res])
    ms, mp_, mc = merge_topk_reference(gs, gp)
    assert torch.equal(res[0]["pids"], mp_) and torch.equal(res[0]["scores"], ms) and torch.equal(res[0]["counts"], mc)
    assert bool((res[0]["scores"][:, :-1] >= res[0]["scores"][:, 1:]).all())
    # and it contains the single-index reference's best document for these queries
    z = dict(np.load(os.path.join(ROOT, "tests", "golden", "idx_nb2.npz")))
    for qi, rec in enumerate(("rank0", "rank3")):
        assert int(z[f"{rec}.final_pids"][0]) in res[0]["pids"][qi].tolist()
