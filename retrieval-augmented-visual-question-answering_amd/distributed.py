"""Passage-sharded search across the GPUs of one node: one process per GPU, each rank owns a contiguous
pid range of the index (IndexArrays.shard / synth.make_corpus(pid_range=...)); centroids, codec tables and queries are
replicated.  The reference has no multi-GPU search at all (src/executors/FLMR_executor.py:778-783 forces the CPU path
when world_size > 1).  Both modes of SURVEY 8e are implemented; the exchanges go through torch.distributed (backend
"nccl" = RCCL over xGMI) or, for a caller without torch, through flmr_topk_allgather / flmr_keys_allreduce of the C ABI:

  exact (`search_batch_exact`, the default of bench.py --gpus N)  three phases with one exchange of u64 keys after each
        (+ an optional phase 0 that splits stage 0 by query): the result is BIT-IDENTICAL to searching the unsharded index;
  fast  (`search_batch`)  every shard runs S0-S4 with the same ndocs and ONE all-gather of the per-shard top-k
        (score f32, global pid i32) lists is merged: exact scores over a superset of the single-index survivors.
"""
import torch
import torch.distributed as dist


def all_gather_topk(scores, pids, group=None):
    """[n, k] per rank -> [world, n, k] on every rank (one fused gather per tensor)."""
    world = dist.get_world_size(group)
    gs = torch.empty((world,) + tuple(scores.shape), dtype=scores.dtype, device=scores.device)
    gp = torch.empty((world,) + tuple(pids.shape), dtype=pids.dtype, device=pids.device)
    try:
        dist.all_gather_into_tensor(gs, scores.contiguous(), group=group)
        dist.all_gather_into_tensor(gp, pids.contiguous(), group=group)
    except (RuntimeError, NotImplementedError):  # backends without the fused form (older gloo)
        dist.all_gather(list(gs.unbind(0)), scores.contiguous(), group=group)
        dist.all_gather(list(gp.unbind(0)), pids.contiguous(), group=group)
    return gs, gp


class ShardedSearcher:
    """local_search(Q, k) -> (pids [n,k] GLOBAL ids, -1 padded; scores [n,k]; counts [n]);  merge(scores, pids) ->
    (scores, pids, counts).  Defaults: the HIP IndexScorer on this rank's shard and the HIP merge kernel."""

    def __init__(self, scorer=None, k_policy=None, local_search=None, merge=None, group=None, topn_keys=None,
                 unpack_keys=None):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.scorer = scorer
        self.k_policy = k_policy or (lambda k: (2, 0.45, 1024) if k <= 100 else (4, 0.4, max(4 * k, 4096)))
        self._local = local_search or self._hip_local_search
        if merge is None:
            from . import ops
            merge = ops.merge_topk
        self._merge = merge
        # key selection / unpacking of the exact protocol: the HIP ops by default (tests on CPU inject numpy restatements)
        self._topn_keys, self._unpack_keys = topn_keys, unpack_keys
        self._split_ok = {}  # (nq, k, nq_cand, options epoch) -> query-split stage 0 usable? (capability query, agreed across ranks once)
        self.timings = None  # set to a dict to collect per-exchange wall times (bench.py --gpus N breakdown)
        self.force_collectives = False  # True: issue the collectives also when world == 1 (bench.py --force-distributed)

    @classmethod
    def from_arrays(cls, arrays, group=None, max_batch=256):
        from .index import DeviceIndex
        from .scorer import IndexScorer
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        shard = arrays.shard(rank, world)
        return cls(scorer=IndexScorer(arrays=shard, device_index=DeviceIndex(shard), max_batch=max_batch), group=group)

    def _hip_local_search(self, Q, k, nq_cand=32, q_lens=None):
        ncells, thr, ndocs = self.k_policy(k)
        return self.scorer.search_batch(Q, k, ncells, thr, ndocs, nq_cand, q_lens=q_lens)

    def search_batch_exact(self, Q, k, nq_cand=32, q_lens=None, gather=None, split_stage0=True, reduce_sum=None, check=True):
        """Exact-parity mode: three phases with one exchange of u64 keys after each; the result is
        bit-identical to searching the unsharded index.  `gather(t)` must return the [world, ...] stack of `t` over the
        ranks (default: torch.distributed.all_gather_into_tensor on the device).  The phase-2/3 outputs are slot-aligned
        with the global survivor list (one non-zero contributor per slot), so they are combined by a SUM all-reduce --
        2(W-1)/W of the array per rank instead of W-1 copies; `reduce_sum(t)` overrides it (default: dist.all_reduce).

        check=True (default): after the batch every rank reads its deferred device status (flmr_searcher_check: candidate
        overflow, q_lens out of range) and the ranks MAX-reduce one error flag, so that a failure on one shard raises on
        EVERY rank instead of leaving the others blocked in the next collective.  It costs a host sync and one tiny
        exchange per batch; a throughput loop passes check=False and calls `check_all()` at its own sync points."""
        if self._topn_keys is None or self._unpack_keys is None:
            from . import ops
            self._topn_keys = self._topn_keys or ops.topn_keys
            self._unpack_keys = self._unpack_keys or ops.unpack_keys
        topn_keys, unpack_keys = self._topn_keys, self._unpack_keys
        ncells, thr, ndocs = self.k_policy(k)
        tm = self.timings

        def timed(name, fn, *a):
            """Run one exchange; with timings enabled, bracket it with device syncs and add its wall time."""
            if tm is None:
                return fn(*a)
            import time
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn(*a)
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            tm[name] = tm.get(name, 0.0) + (time.perf_counter() - t0)
            return out

        def default_gather(t):
            if self.world == 1 and not self.force_collectives:
                return t.unsqueeze(0)
            # flat in / flat out: the one layout both RCCL and gloo accept for the fused gather
            g = torch.empty(self.world * t.numel(), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(g, t.contiguous().view(-1), group=self.group)
            return g.view((self.world,) + tuple(t.shape))

        def default_reduce(t):
            if self.world > 1 or self.force_collectives:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            return t

        if reduce_sum is None and gather is not None:   # a custom gather (tests, host staging): reduce through it
            reduce_sum = lambda t: gather(t).sum(dim=0)
        reduce_sum = reduce_sum or default_reduce
        gather = gather or default_gather

        # Every rank derives the global lists from the same gathered data, and the phase-2/3 outputs are slot-aligned with
        # them: the lists must come out in the SAME ORDER on every rank.  The radix select places its output by a block scan
        # (no atomics), so its order is a function of the input alone and no sort is needed.
        def exchange(keys, n):  # [B, m] per rank -> global top-n per query (reproducible order)
            g = timed("gather_stage1_keys", gather, keys)                 # [W, B, m]
            return topn_keys(g.permute(1, 0, 2).reshape(g.size(1), -1), n, ordered=False)

        k1 = None
        if (self.world > 1 or self.force_collectives) and split_stage0 and self._use_query_split(Q, k, ncells, thr, ndocs, nq_cand, gather):
            # stage 0 does not depend on the passage shard: each rank probes 1/W of the queries, the ranks exchange the
            # idx bitsets + cells (K/8 bytes + a few ints per query) and rebuild the table rows they need locally.
            # Errors raised in here are real failures (the shape was declared supported): they propagate.
            B = Q.size(0)
            per = -(-B // self.world)
            lo = min(B, self.rank * per)
            cnt = min(B, lo + per) - lo
            iw, mc = self.scorer.probe_dims(Q, k, ncells, thr, ndocs, nq_cand)
            dev = self.scorer.probe_device if hasattr(self.scorer, "probe_device") else "cuda"
            bufs = (torch.zeros((per, iw), dtype=torch.int32, device=dev),
                    torch.zeros((per, mc), dtype=torch.int32, device=dev),
                    torch.zeros((per,), dtype=torch.int32, device=dev))
            self.scorer.probe(Q, k, ncells, thr, ndocs, lo, cnt, nq_cand, q_lens=q_lens, out=bufs)
            bits, cells, ncell = (timed("gather_probe_state", gather, t).reshape((-1,) + tuple(t.shape[1:])) for t in bufs)
            k1 = self.scorer.phase1_probed(Q, k, ncells, thr, ndocs, bits, cells, ncell, nq_cand, q_lens=q_lens)
        if k1 is None:
            k1 = self.scorer.phase1(Q, k, ncells, thr, ndocs, nq_cand, q_lens=q_lens)
        s1 = exchange(k1, ndocs)
        s2 = topn_keys(timed("reduce_stage2_keys", reduce_sum, self.scorer.phase2(s1)), ndocs // 4, ordered=False)
        fin = topn_keys(timed("reduce_stage3_keys", reduce_sum, self.scorer.phase3(s2)), min(k, max(ndocs // 4, 1)), ordered=True)
        out = unpack_keys(fin, k)
        if check:
            self.check_all(gather)
        return out

    def check_all(self, gather=None):
        """Collective: every rank reads its searcher's deferred status (waits for its last batch) and the ranks exchange one
        flag; raises on ALL ranks if any shard failed (the failing rank re-raises its own error, the others name the rank)."""
        err = None
        try:
            if hasattr(self.scorer, "check"):
                self.scorer.check()
        except Exception as e:  # noqa: BLE001 -- whatever the shard raised must reach the other ranks as a flag
            err = e
        dev = self.scorer.probe_device if hasattr(self.scorer, "probe_device") else "cuda"
        flag = torch.tensor([1 if err is not None else 0], dtype=torch.int32, device=dev)
        if gather is not None:
            flags = gather(flag).reshape(-1)
        elif self.world > 1 or self.force_collectives:
            flags = torch.empty(self.world, dtype=torch.int32, device=dev)
            dist.all_gather_into_tensor(flags, flag, group=self.group)
        else:
            flags = flag
        if err is not None:
            raise err
        bad = torch.nonzero(flags).reshape(-1).tolist()
        if bad:
            raise RuntimeError(f"sharded search failed on rank(s) {bad} (see that rank's error)")

    def _use_query_split(self, Q, k, ncells, thr, ndocs, nq_cand, gather):
        """Capability query (flmr_searcher_probe_supported: depends only on replicated data), then -- once per batch shape
        -- the MIN of the answers over the ranks, so that a rank can never take a different branch of the protocol (and
        issue different collectives) than the others."""
        from . import _native
        # the answer also depends on the kernel switches (flmr_set_option) and on whether the scorer keeps the full table
        key = (int(Q.size(1)), int(k), int(nq_cand), _native.options_epoch, bool(getattr(self.scorer, "full_table_state", False)))
        if key not in self._split_ok:
            ok = bool(self.scorer.supports_query_split(Q, k, ncells, thr, ndocs, nq_cand))
            dev = self.scorer.probe_device if hasattr(self.scorer, "probe_device") else "cuda"
            votes = gather(torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev))
            self._split_ok[key] = bool(int(votes.min()) == 1)
        return self._split_ok[key]

    def search_batch(self, Q, k, **kw):
        pids, scores, counts = self._local(Q, k, **kw)
        if self.world == 1:
            return pids, scores, counts
        gs, gp = all_gather_topk(scores, pids, self.group)
        ms, mp, mc = self._merge(gs, gp)
        return mp, ms, mc
