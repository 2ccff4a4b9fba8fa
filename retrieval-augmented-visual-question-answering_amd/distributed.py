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


class _Pending:
    """An exchange that has been issued: `wait()` makes the current stream wait for it (device collectives) and returns its
    result.  `keep` holds the send buffer alive until then."""
    __slots__ = ("out", "work", "keep")

    def __init__(self, out, work=None, keep=None):
        self.out, self.work, self.keep = out, work, keep

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = self.keep = None
        return self.out


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

    def _hip_local_search(self, Q, k, nq_cand=32, q_lens=None, checked=False):
        """This shard's search, queued on the current stream (no host sync: the all-gather of the fast mode follows on the same
        stream).  checked=True reads the deferred device status after the batch (a host sync: candidate overflow, q_lens out of
        range -- nothing recoverable is left for the host to do: a query over the score-row capacity is handled inside the
        library); a throughput loop leaves it off and calls `check_all()` at its own sync points."""
        ncells, thr, ndocs = self.k_policy(k)
        if checked and hasattr(self.scorer, "search_batch_checked"):
            return self.scorer.search_batch_checked(Q, k, ncells, thr, ndocs, nq_cand, q_lens=q_lens)
        return self.scorer.search_batch(Q, k, ncells, thr, ndocs, nq_cand, q_lens=q_lens)

    # ---- the exact protocol as a coroutine: it YIELDS every exchange it has issued (a _Pending) and is resumed with the exchanged
    # data, so that a driver can run one batch to completion (search_batch_exact) or interleave the sub-batches of a step
    # (search_batch_exact_pipelined: the exchange of one sub-batch travels while the next one computes) --------------------------
    def _exchangers(self, gather, reduce_sum):
        """-> (start_gather(name, t) -> _Pending of the [world, ...] stack, start_reduce(name, t) -> _Pending of the SUM).
        Default collectives are issued with async_op=True on the device (RCCL's own stream; `wait()` makes the CURRENT stream
        wait, never the host); injected callables (tests, host staging) run synchronously.  With `self.timings` set every
        exchange is instead run synchronously between two events on the launch stream and its duration recorded."""
        tm = self.timings
        use_cuda = torch.cuda.is_available()

        def timed(name, fn, t):
            import time
            if use_cuda and t.is_cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = fn(t)
                e1.record()
                tm.setdefault("_events", []).append((name, e0, e1))
                return out
            t0 = time.perf_counter()
            out = fn(t)
            tm[name] = tm.get(name, 0.0) + (time.perf_counter() - t0)
            return out

        def sync_gather(t):
            if gather is not None:
                return gather(t)
            if self.world == 1 and not self.force_collectives:
                return t.unsqueeze(0)
            # flat in / flat out: the one layout both RCCL and gloo accept for the fused gather
            g = torch.empty(self.world * t.numel(), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(g, t.contiguous().view(-1), group=self.group)
            return g.view((self.world,) + tuple(t.shape))

        def sync_reduce(t):
            if reduce_sum is not None:
                return reduce_sum(t)
            if gather is not None:   # a custom gather (tests, host staging): reduce through it
                return gather(t).sum(dim=0)
            if self.world > 1 or self.force_collectives:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            return t

        def start_gather(name, t):
            if tm is not None:
                return _Pending(timed(name, sync_gather, t))
            if gather is not None or (self.world == 1 and not self.force_collectives):
                return _Pending(sync_gather(t))
            src = t.contiguous().view(-1)
            g = torch.empty(self.world * t.numel(), dtype=t.dtype, device=t.device)
            work = dist.all_gather_into_tensor(g, src, group=self.group, async_op=True)
            return _Pending(g.view((self.world,) + tuple(t.shape)), work, keep=src)

        def start_reduce(name, t):
            if tm is not None:
                return _Pending(timed(name, sync_reduce, t))
            if reduce_sum is not None or gather is not None or not (self.world > 1 or self.force_collectives):
                return _Pending(sync_reduce(t))
            work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            return _Pending(t, work)

        return start_gather, start_reduce, sync_gather

    def _exact_steps(self, scorer, Q, k, nq_cand, q_lens, gather, reduce_sum, split_stage0, truncate_phase1, cert):
        """Coroutine of one batch through the exact protocol on `scorer` (see search_batch_exact).  Yields _Pending objects
        (or lists of them), is resumed with their results; returns (pids, scores, counts)."""
        if self._topn_keys is None or self._unpack_keys is None:
            from . import ops
            self._topn_keys = self._topn_keys or ops.topn_keys
            self._unpack_keys = self._unpack_keys or ops.unpack_keys
        topn_keys, unpack_keys = self._topn_keys, self._unpack_keys
        ncells, thr, ndocs = self.k_policy(k)
        start_gather, start_reduce, sync_gather = self._exchangers(gather, reduce_sum)

        # Every rank derives the global lists from the same gathered data, and the phase-2/3 outputs are slot-aligned with
        # them: the lists must come out in the SAME ORDER on every rank.  The radix select places its output by a block scan
        # (no atomics), so its order is a function of the input alone and no sort is needed.
        k1 = None
        if (self.world > 1 or self.force_collectives) and split_stage0 and self._use_query_split(scorer, Q, k, ncells, thr, ndocs, nq_cand, sync_gather):
            # stage 0 does not depend on the passage shard: each rank probes 1/W of the queries, the ranks exchange the
            # idx bitsets + cells (K/8 bytes + a few ints per query) and rebuild the table rows they need locally.
            # Errors raised in here are real failures (the shape was declared supported): they propagate.
            B = Q.size(0)
            per = -(-B // self.world)
            lo = min(B, self.rank * per)
            cnt = min(B, lo + per) - lo
            iw, mc = scorer.probe_dims(Q, k, ncells, thr, ndocs, nq_cand)
            dev = scorer.probe_device if hasattr(scorer, "probe_device") else "cuda"
            bufs = (torch.zeros((per, iw), dtype=torch.int32, device=dev),
                    torch.zeros((per, mc), dtype=torch.int32, device=dev),
                    torch.zeros((per,), dtype=torch.int32, device=dev))
            scorer.probe(Q, k, ncells, thr, ndocs, lo, cnt, nq_cand, q_lens=q_lens, out=bufs)
            got = yield [start_gather("gather_probe_state", t) for t in bufs]
            bits, cells, ncell = (g.reshape((-1,) + tuple(t.shape[1:])) for g, t in zip(got, bufs))
            k1 = scorer.phase1_probed(Q, k, ncells, thr, ndocs, bits, cells, ncell, nq_cand, q_lens=q_lens)
        if k1 is None:
            k1 = scorer.phase1(Q, k, ncells, thr, ndocs, nq_cand, q_lens=q_lens)
        # [B, ndocs] per rank -> global top-ndocs per query (reproducible order)
        m = phase1_width(ndocs, self.world) if truncate_phase1 else ndocs
        if m < ndocs and k1.size(1) > m:
            local = topn_keys(k1, m, ordered=False)                              # this shard's m best
            g = yield start_gather("gather_stage1_keys", local)                 # [W, B, m]
            s1, violated = merge_truncated(g, ndocs, topn_keys)
            cert.append(violated)
        else:
            g = yield start_gather("gather_stage1_keys", k1)                    # [W, B, ndocs]
            s1 = topn_keys(g.permute(1, 0, 2).reshape(g.size(1), -1), ndocs, ordered=False)
        s2 = topn_keys((yield start_reduce("reduce_stage2_keys", scorer.phase2(s1))), ndocs // 4, ordered=False)
        fin = topn_keys((yield start_reduce("reduce_stage3_keys", scorer.phase3(s2))), min(k, max(ndocs // 4, 1)), ordered=True)
        return unpack_keys(fin, k)

    @staticmethod
    def _wait(p):
        return [x.wait() for x in p] if isinstance(p, list) else p.wait()

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
        cert = []
        gen = self._exact_steps(self.scorer, Q, k, nq_cand, q_lens, gather, reduce_sum, split_stage0, truncate_phase1, cert)
        try:
            p = next(gen)
            while True:
                p = gen.send(self._wait(p))
        except StopIteration as e:
            out = e.value
        if check:
            self.check_all(gather)
            if cert and bool(cert[0]):   # (never on evenly sharded data) redo this batch with the full phase-1 exchange
                return self.search_batch_exact(Q, k, nq_cand=nq_cand, q_lens=q_lens, gather=gather, split_stage0=split_stage0,
                                               reduce_sum=reduce_sum, check=True, truncate_phase1=False)
        else:
            self._cert += cert
        return out

    def search_batch_exact_pipelined(self, Q, k, nq_cand=32, q_lens=None, sub_batch=256, depth=2, gather=None, split_stage0=True,
                                     reduce_sum=None, truncate_phase1=True):
        """The exact protocol over a step's queries cut into sub-batches of `sub_batch`, `depth` of them in flight: while the
        exchange one sub-batch has issued travels (RCCL's stream), the next sub-batch's phase computes (the launch stream) --
        a step no longer waits out four exchanges with an idle device, and a rank's workspace is sized for a sub-batch, not
        for the step.  Each in-flight sub-batch has its own native searcher (the phases of a batch share workspace state);
        the schedule -- which sub-batch issues which collective when -- is a function of the batch shape alone, so every rank
        issues the same collectives in the same order.  Queries are independent: the result is bit-identical to
        search_batch_exact on the whole batch (tests).  Unchecked like check=False: call `check_all()` at a sync point."""
        B = Q.size(0)
        chunks = [(lo, min(B, lo + sub_batch)) for lo in range(0, B, sub_batch)]
        if len(chunks) <= 1 or depth <= 1:
            return self.search_batch_exact(Q, k, nq_cand=nq_cand, q_lens=q_lens, gather=gather, split_stage0=split_stage0,
                                           reduce_sum=reduce_sum, check=False, truncate_phase1=truncate_phase1)
        scorers = self._pipeline_scorers(min(depth, len(chunks)))
        cert, results = [], [None] * len(chunks)

        def start(ci, slot):
            lo, hi = chunks[ci]
            gen = self._exact_steps(scorers[slot], Q[lo:hi], k, nq_cand, None if q_lens is None else q_lens[lo:hi], gather, reduce_sum,
                                    split_stage0, truncate_phase1, cert)
            return [gen, next(gen), ci]

        active = [start(ci, ci) for ci in range(len(scorers))]
        nxt = len(active)
        while any(a is not None for a in active):
            for slot, a in enumerate(active):
                if a is None:
                    continue
                gen, pend, ci = a
                try:
                    a[1] = gen.send(self._wait(pend))
                except StopIteration as e:
                    results[ci] = e.value
                    if nxt < len(chunks):
                        active[slot] = start(nxt, slot)
                        nxt += 1
                    else:
                        active[slot] = None
        self._cert += cert
        return tuple(torch.cat([r[j] for r in results]) for j in range(3))

    def _pipeline_scorers(self, n):
        """The scorers of the in-flight sub-batches: this rank's scorer + clones on the same resident index."""
        if not hasattr(self, "_pipe"):
            self._pipe = [self.scorer]
        while len(self._pipe) < n:
            self._pipe.append(self.scorer.clone())
        return self._pipe[:n]

    def exchange_ms(self, reset=True):
        """{exchange name: milliseconds} accumulated since `timings` was set: event-timed on the launch stream for device tensors
        (no host sync inside the protocol), wall time for host tensors."""
        tm = self.timings or {}
        out = {n: v * 1e3 for n, v in tm.items() if n != "_events"}
        evs = tm.get("_events", [])
        if evs:
            torch.cuda.synchronize()
            for name, e0, e1 in evs:
                out[name] = out.get(name, 0.0) + e0.elapsed_time(e1)
        if reset and self.timings is not None:
            self.timings = {}
        return out

    def check_all(self, gather=None):
        """Collective: every rank reads its searchers' deferred status (waits for their last batches) and the ranks exchange one
        flag; raises on ALL ranks if any shard failed (the failing rank re-raises its own error, the others name the rank).
        EVERY scorer of the pipeline is polled (and its flags cleared) before the first error is kept: a clone left with a stale
        flag would fail its next phase call on this rank only, with the other ranks already inside the collective."""
        err = None
        for sc in getattr(self, "_pipe", [self.scorer]):
            if not hasattr(sc, "check"):
                continue
            try:
                sc.check()
            except Exception as e:  # noqa: BLE001 -- whatever the shard raised must reach the other ranks as a flag
                err = err or e
        dev = self.scorer.probe_device if hasattr(self.scorer, "probe_device") else "cuda"
        flag = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=dev)
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

    def _use_query_split(self, scorer, Q, k, ncells, thr, ndocs, nq_cand, gather):
        """Capability query (flmr_searcher_probe_supported: depends only on replicated data), then -- once per batch shape
        -- the MIN of the answers over the ranks, so that a rank can never take a different branch of the protocol (and
        issue different collectives) than the others."""
        from . import _native
        # the answer also depends on the kernel switches (flmr_set_option) and on whether the scorer keeps the full table
        key = (int(Q.size(1)), int(k), int(nq_cand), _native.options_epoch, bool(getattr(scorer, "full_table_state", False)))   # (full table: set by taps / retrieve())
        if key not in self._split_ok:
            ok = bool(scorer.supports_query_split(Q, k, ncells, thr, ndocs, nq_cand))
            dev = scorer.probe_device if hasattr(scorer, "probe_device") else "cuda"
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
