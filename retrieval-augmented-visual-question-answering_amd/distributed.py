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


_SIGN = -(1 << 63)   # keys are u64 bit patterns held in int64 tensors: x ^ _SIGN maps unsigned order to signed order (0 -> smallest)


def phase1_width(ndocs, world, slack=64):
    """Keys per shard in the truncated phase-1 exchange: twice the shard's expected share of the global top-ndocs plus slack
    (a multiple of 64), never more than ndocs.  A passage-sharded index holds ~ndocs / world of the global survivors per
    shard (binomial: sd < sqrt(ndocs / world)), so the certificate below fails only on adversarially skewed shards."""
    m = 2 * (-(-ndocs // world)) + slack
    m = -(-m // 64) * 64
    return min(ndocs, m)


def merge_truncated(g, n, topn_keys):
    """g [W, B, m]: every shard's m best stage-1 keys (unordered, 0 = empty).  -> (the n best of their union [B, n], unordered;
    violated: 0-d bool tensor).  The union's top-n equals the top-n over ALL the shards' keys iff no shard omitted a key that
    belongs to it.  A shard that sent fewer than m keys omitted nothing; one that sent m omitted only keys BELOW its smallest
    sent key, so it is enough that this smallest key did not itself make the top-n (it is below the n-th kept key).  Every
    rank evaluates this on the same gathered bytes, so all ranks see the same verdict."""
    W, B, m = g.shape
    out = topn_keys(g.permute(1, 0, 2).reshape(B, W * m), n, ordered=False)
    kept_min = (out ^ _SIGN).min(dim=1).values                    # n-th kept key in signed order (smallest if fewer than n exist)
    shard_min = (g ^ _SIGN).min(dim=2).values                     # [W, B]; a shard with an empty slot maps to the smallest value
    full = (g != 0).all(dim=2)
    violated = (full & (shard_min >= kept_min.unsqueeze(0))).any()
    return out, violated


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
        self._cert = []                 # device flags of unchecked batches: truncated phase-1 exchange certificate violated?

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

    def search_batch_exact(self, Q, k, nq_cand=32, q_lens=None, gather=None, split_stage0=True, reduce_sum=None, check=True,
                           truncate_phase1=True):
        """Exact-parity mode: three phases with one exchange of u64 keys after each; the result is
        bit-identical to searching the unsharded index.  `gather(t)` must return the [world, ...] stack of `t` over the
        ranks (default: torch.distributed.all_gather_into_tensor on the device).  The phase-2/3 outputs are slot-aligned
        with the global survivor list (one non-zero contributor per slot), so they are combined by a SUM all-reduce --
        2(W-1)/W of the array per rank instead of W-1 copies; `reduce_sum(t)` overrides it (default: dist.all_reduce).

        check=True (default): after the batch every rank reads its deferred device status (flmr_searcher_check: candidate
        overflow, q_lens out of range) and the ranks MAX-reduce one error flag, so that a failure on one shard raises on
        EVERY rank instead of leaving the others blocked in the next collective.  It costs a host sync and one tiny
        exchange per batch; a throughput loop passes check=False and calls `check_all()` at its own sync points.

        truncate_phase1=True: after phase 1 a shard ships only its `phase1_width(ndocs, world)` best keys instead of all ndocs
        (8 MB -> 2.5 MB per rank and 1024 queries at 8 shards) together with a certificate evaluated on the gathered data
        (`merge_truncated`).  The certificate's verdict stays on the device: with check=True it is read at the batch's sync
        point and a violated batch is redone with the full exchange (the result is exact either way); with check=False it is
        read by `check_all()`, which raises."""
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
        cert = []

        def exchange(keys, n):  # [B, ndocs] per rank -> global top-n per query (reproducible order)
            m = phase1_width(n, self.world) if truncate_phase1 else n
            if m < n and keys.size(1) > m:
                local = topn_keys(keys, m, ordered=False)                  # this shard's m best
                g = timed("gather_stage1_keys", gather, local)            # [W, B, m]
                out, violated = merge_truncated(g, n, topn_keys)
                cert.append(violated)
                return out
            g = timed("gather_stage1_keys", gather, keys)                 # [W, B, ndocs]
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
            if cert and bool(cert[0]):   # (never on evenly sharded data) redo this batch with the full phase-1 exchange
                return self.search_batch_exact(Q, k, nq_cand=nq_cand, q_lens=q_lens, gather=gather, split_stage0=split_stage0,
                                               reduce_sum=reduce_sum, check=True, truncate_phase1=False)
        else:
            self._cert += cert
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
        pending, self._cert = self._cert, []
        if err is not None:
            raise err
        bad = torch.nonzero(flags).reshape(-1).tolist()
        if bad:
            raise RuntimeError(f"sharded search failed on rank(s) {bad} (see that rank's error)")
        if pending and bool(torch.stack([c.reshape(()) for c in pending]).any()):   # identical on every rank (same gathered data)
            raise RuntimeError("truncated phase-1 exchange: a shard held more of the global stage-1 survivors than it shipped "
                               "(skewed sharding); the unchecked batches since the last check_all() are not exact -- rerun them "
                               "with truncate_phase1=False or check=True")

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
