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


# ---- exact-parity mode: three phases, real collectives, 2 and 3 ranks ------------------------------------------
def _exact_worker(rank, world, port, out_path, query_split, use_q_lens):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ravqa_amd
    from ravqa_amd.distributed import ShardedSearcher
    import oracle_shard_scorer as oss
    z = dict(np.load(os.path.join(ROOT, "tests", "golden", "idx_nb2.npz")))
    full = ravqa_amd.IndexArrays.from_golden(z)
    scorer = oss.OracleShardScorer(full.shard(rank, world), query_split=query_split)
    policy = {100: (2, 0.45, 1024), 10: (1, 0.5, 64)}
    ss = ShardedSearcher(scorer=scorer, k_policy=lambda k: policy[k], topn_keys=oss.topn_keys, unpack_keys=oss.unpack_keys)
    ss.timings = {}
    # 5 queries: with 2 ranks the slices are 3 + 2, with 3 ranks 2 + 2 + 1 (ragged last slice)
    Q = torch.stack([torch.from_numpy(z[f"rank{i}.Q"]) for i in (0, 3, 1, 4, 2)])
    q_lens = torch.tensor([32, 20, 32, 7, 32], dtype=torch.int32) if use_q_lens else None
    res = {}
    for k in (100, 10):
        p, s, c = ss.search_batch_exact(Q, k, nq_cand=32, q_lens=q_lens)   # default gather / all_reduce: real gloo collectives
        res[k] = (p, s, c)
    timings = dict(ss.timings)
    # the same step cut into sub-batches with two of them in flight (search_batch_exact_pipelined): asynchronous collectives,
    # one scorer per in-flight sub-batch, ragged last sub-batch -- must equal the whole-batch call bit for bit
    ss.timings = None
    Q7 = torch.cat([Q, Q[[2, 0]]])
    ql7 = None if q_lens is None else torch.cat([q_lens, q_lens[[2, 0]]])
    whole = ss.search_batch_exact(Q7, 100, nq_cand=32, q_lens=ql7)
    piped = {sb: ss.search_batch_exact_pipelined(Q7, 100, nq_cand=32, q_lens=ql7, sub_batch=sb, depth=dp) for sb, dp in ((3, 2), (2, 3))}
    ss.check_all()
    torch.save({"res": res, "calls": scorer.calls, "timings": timings, "split_ok": dict(ss._split_ok), "whole": whole, "piped": piped},
               f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,query_split,use_q_lens", [(2, True, False), (3, True, True), (2, False, True), (4, True, True), (8, True, False)])
def test_exact_protocol_multirank_equals_unsharded(tmp_path, world, query_split, use_q_lens):
    """ShardedSearcher.search_batch_exact over real gloo collectives (all_gather_into_tensor of the probe state and of the
    stage-1 keys, SUM all-reduce of the slot-aligned stage-2/3 keys): every rank ends with the ranking of the UNSHARDED
    index, bit for bit -- ids, scores, counts -- for both k-policies, with a ragged last query slice, with per-query
    lengths, and with the replicated-stage-0 fallback (capability vote = no)."""
    from oracle import oracle as orc
    import ravqa_amd
    port, out = _free_port(), str(tmp_path / "res")
    mp.spawn(_exact_worker, args=(world, port, out, query_split, use_q_lens), nprocs=world, join=True)
    res = [torch.load(f"{out}.{r}") for r in range(world)]
    z = dict(np.load(os.path.join(ROOT, "tests", "golden", "idx_nb2.npz")))
    oi = orc.OracleIndex.from_golden(z)
    Q = np.stack([z[f"rank{i}.Q"] for i in (0, 3, 1, 4, 2)])
    q_lens = [32, 20, 32, 7, 32] if use_q_lens else [32] * 5
    for k, (ncells, thr, ndocs) in {100: (2, 0.45, 1024), 10: (1, 0.5, 64)}.items():
        for r in range(1, world):   # all ranks agree
            for a, b in zip(res[0]["res"][k], res[r]["res"][k]):
                assert torch.equal(a, b), (k, r)
        p, s, c = res[0]["res"][k]
        for b in range(5):
            rp, rs, ncand = oi.rank(Q[b][: q_lens[b]], ncells, thr, ndocs, 32)
            n = int(c[b])
            assert n == min(k, len(rp)), (k, b, n, len(rp))
            if ncand < ndocs:
                continue   # the reference's undefined case (filter_pids.cpp:119-123); defined here, covered elsewhere
            assert p[b, :n].tolist() == rp[:n].tolist(), (k, b)
            assert np.array_equal(s[b, :n].numpy().view(np.uint32), rs[:n].view(np.uint32)), (k, b)
    for r in range(world):
        calls = [c[0] for c in res[r]["calls"]]
        if query_split:
            per = -(-5 // world)
            lo = min(5, r * per)
            assert ("probe", lo, min(5, lo + per) - lo) in res[r]["calls"] and "phase1_probed" in calls and "phase1" not in calls
            assert "gather_probe_state" in res[r]["timings"]
        else:
            assert "phase1" in calls and "probe" not in calls and "gather_probe_state" not in res[r]["timings"]
        assert list(res[r]["split_ok"].values()) == [query_split] * len(res[r]["split_ok"])
        assert {"gather_stage1_keys", "reduce_stage2_keys", "reduce_stage3_keys"} <= set(res[r]["timings"])
        for sb, got in res[r]["piped"].items():   # pipelined sub-batches == one call, on every rank
            for a, b in zip(got, res[r]["whole"]):
                assert torch.equal(a, b), (r, sb)


def _vote_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ravqa_amd
    from ravqa_amd.distributed import ShardedSearcher
    import oracle_shard_scorer as oss
    z = dict(np.load(os.path.join(ROOT, "tests", "golden", "idx_nb2.npz")))
    full = ravqa_amd.IndexArrays.from_golden(z)
    scorer = oss.OracleShardScorer(full.shard(rank, world), query_split=(rank == 0))   # the ranks DISAGREE
    ss = ShardedSearcher(scorer=scorer, k_policy=lambda k: (1, 0.5, 64), topn_keys=oss.topn_keys, unpack_keys=oss.unpack_keys)
    Q = torch.stack([torch.from_numpy(z[f"rank{i}.Q"]) for i in (0, 3)])
    p, s, c = ss.search_batch_exact(Q, 10)
    torch.save({"p": p, "calls": [c_[0] for c_ in scorer.calls]}, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_that_disagree_on_the_capability_take_the_same_branch(tmp_path):
    """One rank says the query-split stage 0 is supported, the other does not: the MIN vote sends BOTH down the replicated
    branch (no probe anywhere, identical results) instead of issuing mismatched collectives."""
    world, port, out = 2, _free_port(), str(tmp_path / "res")
    mp.spawn(_vote_worker, args=(world, port, out), nprocs=world, join=True)
    res = [torch.load(f"{out}.{r}") for r in range(world)]
    assert torch.equal(res[0]["p"], res[1]["p"])
    for r in res:
        assert "probe" not in r["calls"] and "phase1_probed" not in r["calls"] and "phase1" in r["calls"]


def _failing_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ravqa_amd
    from ravqa_amd.distributed import ShardedSearcher
    import oracle_shard_scorer as oss
    z = dict(np.load(os.path.join(ROOT, "tests", "golden", "idx_nb2.npz")))
    full = ravqa_amd.IndexArrays.from_golden(z)
    scorer = oss.OracleShardScorer(full.shard(rank, world))
    if rank == 1:   # this shard's deferred device status reports a failure (flmr_searcher_check's role)
        def check():
            raise ravqa_amd._native.FlmrNativeError("libflmr_hip status 5: candidate capacity exceeded (injected)")
        scorer.check = check
    ss = ShardedSearcher(scorer=scorer, k_policy=lambda k: (1, 0.5, 64), topn_keys=oss.topn_keys, unpack_keys=oss.unpack_keys)
    Q = torch.stack([torch.from_numpy(z[f"rank{i}.Q"]) for i in (0, 3)])
    msg = "no error"
    try:
        ss.search_batch_exact(Q, 10)
    except Exception as e:  # noqa: BLE001
        msg = f"{type(e).__name__}: {e}"
    # unchecked batches do not exchange the flag; check_all() at a sync point of the caller's choosing does
    ss.search_batch_exact(Q, 10, check=False)
    msg2 = "no error"
    try:
        ss.check_all()
    except Exception as e:  # noqa: BLE001
        msg2 = f"{type(e).__name__}: {e}"
    torch.save({"msg": msg, "msg2": msg2}, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_failure_on_one_shard_raises_on_every_rank(tmp_path):
    """A shard whose deferred device status reports an error (candidate overflow, bad q_lens) must not leave the other ranks
    blocked in the next collective: search_batch_exact(check=True) exchanges one flag per batch and EVERY rank raises."""
    world, port, out = 3, _free_port(), str(tmp_path / "res")
    mp.spawn(_failing_worker, args=(world, port, out), nprocs=world, join=True)
    res = [torch.load(f"{out}.{r}") for r in range(world)]
    for key in ("msg", "msg2"):
        assert "FlmrNativeError" in res[1][key] and "injected" in res[1][key]
        for r in (0, 2):
            assert "RuntimeError" in res[r][key] and "rank(s) [1]" in res[r][key], res[r][key]


def test_truncated_phase1_exchange_certificate():
    """merge_truncated (the phase-1 exchange that ships each shard's 2 * ndocs / W + slack best keys): on evenly spread keys
    the union's top-n equals the top-n of all keys and the certificate holds; when one shard holds more of the global top-n
    than it shipped the certificate FAILS (so the caller redoes the batch with the full exchange) -- it must never pass a
    wrong result."""
    import oracle_shard_scorer as oss
    from ravqa_amd.distributed import merge_truncated, phase1_width
    rng = np.random.default_rng(0)
    W, B, n = 8, 5, 1024
    m = phase1_width(n, W)
    assert m == 320 and phase1_width(n, 2) == n and phase1_width(64, 3) == 64
    u = lambda a: torch.from_numpy(a.astype(np.uint64).view(np.int64))
    sel = lambda keys, k: np.sort(np.ascontiguousarray(keys).view(np.uint64), axis=-1)[..., ::-1][..., :k]
    for skew in (False, True):
        # per shard 600 keys (high word = score order incl. "negative" scores below 2^63, low word = pid)
        allk = rng.integers(1, 1 << 62, size=(W, B, 600), dtype=np.int64).astype(np.uint64) * np.uint64(3)
        if skew:
            allk[0] |= np.uint64(1) << np.uint64(63)        # shard 0 holds every large key of every query
        g = np.zeros((W, B, m), dtype=np.uint64)
        for w in range(W):
            g[w] = sel(allk[w], m)
        out, violated = merge_truncated(u(g), n, oss.topn_keys)
        truth = sel(allk.transpose(1, 0, 2).reshape(B, -1), n)
        same = np.array_equal(np.sort(out.numpy().view(np.uint64), axis=1)[:, ::-1], truth)
        assert bool(violated) == skew and same == (not skew)
    # a shard with fewer keys than the width omitted nothing: no violation although all its keys are kept
    g = np.zeros((W, B, m), dtype=np.uint64)
    g[:, :, :100] = rng.integers(1, 1 << 62, size=(W, B, 100), dtype=np.int64).astype(np.uint64)
    out, violated = merge_truncated(u(g), n, oss.topn_keys)
    assert not bool(violated) and int((out != 0).sum()) == B * W * 100
