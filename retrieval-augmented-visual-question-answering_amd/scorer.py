"""`IndexScorer` -- host-side mirror of TPC/search/index_storage.py:21-182 (IndexLoader + CandidateGeneration +
scoring) driving the HIP search path through the C ABI.

Device policy: the reference reads `use_gpu=False` as "CPU extensions, fp32, zero-clamped
MaxSim".  This build ALWAYS runs on the MI355X with exactly those CPU-path numerics, whatever `use_gpu`
says; there is no host fallback (a missing library / device raises FlmrNativeError).
"""
import ctypes as C

import numpy as np
import torch

from . import _native
from .index import DeviceIndex, IndexArrays, codec_tables, load_index_arrays


class _Codec:
    """The attributes of ResidualCodec that callers of IndexScorer read (residual.py:19-95)."""

    def __init__(self, arrays: IndexArrays):
        self.dim, self.nbits = arrays.dim, arrays.nbits
        self.centroids = torch.from_numpy(arrays.centroids)
        self.bucket_weights = torch.from_numpy(arrays.bucket_weights)
        self.bucket_cutoffs = None if arrays.bucket_cutoffs is None else torch.from_numpy(np.asarray(arrays.bucket_cutoffs))
        self.avg_residual = arrays.avg_residual
        rev, lut = codec_tables(arrays.nbits)
        self.reversed_bit_map = torch.from_numpy(rev)
        self.decompression_lookup_table = torch.from_numpy(lut)
        self.use_gpu = True


class _Embeddings:
    def __init__(self, arrays):
        self.codes = torch.from_numpy(arrays.codes)
        self.residuals = torch.from_numpy(arrays.residuals)


class _Strided:
    """Packed ragged container: `.tensor`, `.lengths`, `.offsets` (strided_tensor_core.py:17-31)."""

    def __init__(self, tensor, lengths):
        self.tensor = tensor
        self.lengths = lengths.long()
        self.offsets = torch.cat((torch.zeros(1, dtype=torch.long), torch.cumsum(self.lengths, dim=0)))


class _EmbeddingsStrided:
    def __init__(self, emb, doclens):
        self.codes_strided = _Strided(emb.codes, doclens)
        self.residuals_strided = _Strided(emb.residuals, doclens)


def _op(name):
    def call(*args, **kw):
        from . import ops
        return getattr(ops, name)(*args, **kw)
    call.__name__ = name
    return staticmethod(call)


def _params(k, ncells, thr, ndocs, nq_cand):
    return _native.SearchParams(int(k), int(ncells), float(thr), int(ndocs), int(nq_cand))


class PendingBatch:
    """Results of one batched search on their way to pinned host memory (IndexScorer.search_batch_pending): numpy views `pids`
    [n, k], `scores` [n, k], `counts` [n] whose rows [b0, b1) are valid once `wait(chunk)` has returned."""

    def __init__(self, hp, hs, hc, marks, device_results, check):
        self._t = (hp, hs, hc)                      # the pinned tensors (kept alive with the views)
        self.pids, self.scores, self.counts = hp.numpy(), hs.numpy(), hc.numpy()
        self._marks, self._dev, self._check = marks, device_results, check
        self._done = [False] * len(marks)

    @classmethod
    def resolved(cls, pids, scores, counts):
        self = cls.__new__(cls)
        self._t = (pids.cpu(), scores.cpu(), counts.cpu())
        self.pids, self.scores, self.counts = (t.numpy() for t in self._t)
        self._marks, self._dev, self._check = [(0, self.pids.shape[0], None, None)], None, None
        self._done = [True]
        return self

    def chunks(self):
        return [(b0, b1) for b0, b1, _, _ in self._marks]

    def wait(self, j):
        if self._done[j]:
            return
        _, _, ev, flags = self._marks[j]
        ev.synchronize()
        if int(flags[0]) or int(flags[1]):
            self._check()   # a deferred device-side error up to this sub-batch (candidate bound, q_lens range): raises it
        self._done[j] = True

    def wait_all(self):
        for j in range(len(self._done)):
            self.wait(j)


class IndexScorer:
    # class attributes the reference installs from its JIT-built extensions (index_storage.py:29-60)
    filter_pids = _op("filter_pids")
    decompress_residuals = _op("decompress_residuals")

    NUMERICS = {"cpu": 0, "gpu-fp16": 1}   # FLMR_NUMERICS_CPU / FLMR_NUMERICS_GPU_FP16 (include/flmr_hip.h)

    def __init__(self, index_path=None, use_gpu=True, arrays: IndexArrays = None, device_index: DeviceIndex = None,
                 max_batch=256, streams=1, numerics=None):
        """numerics: "cpu" (default; FLMR_NUMERICS=gpu-fp16 overrides the default) = the reference's CPU-path arithmetic, the
        pinned claim; "gpu-fp16" = the reference's CUDA-path arithmetic (fp16 scores / embeddings, -9999 padding, no
        clamp: index_storage.py:113-158) -- opt-in (see `Searcher`); an index that mode cannot serve (centroids not
        fp16-representable, K % 64 != 0) falls back to "cpu" with a warning.  Either way everything runs on the MI355X.
        max_batch: queries per native call (one workspace holds that many); a larger batch is cut into sub-batches.
        streams: the sub-batches of one search_batch call are dealt round-robin to this many native searchers, each on
        its own HIP stream, joined with the caller's stream at both ends.  Default 1: with the join a stream-ordered call
        needs, two streams measured 10.35 ms against 9.60 ms for the same four sub-batches of 256 in sequence (both run the
        same stage at the same time and compete for it); only free-running streams that drift out of phase gained (9.3 ms,
        profiles/stream_phase_probe.py), which is a caller-level choice (one IndexScorer per stream)."""
        if arrays is None and device_index is None:
            arrays = load_index_arrays(index_path)
        import os
        env = os.environ.get("FLMR_NUMERICS")
        self.numerics = numerics or (env if env in self.NUMERICS else None) or "cpu"   # ("reference" is resolved by Searcher)
        if self.numerics not in self.NUMERICS:
            raise ValueError(f"numerics must be one of {sorted(self.NUMERICS)}, got {self.numerics!r}")
        self.index_path = index_path
        self.use_gpu = True  # see module docstring
        self.arrays = arrays if arrays is not None else device_index.arrays
        self.device_index = device_index or DeviceIndex(self.arrays)
        self._lib = _native.load(require_device=True)
        if self.numerics == "gpu-fp16":
            info = self.device_index.info()
            if not (info["centroids_f16_exact"] and self.arrays.num_centroids % 64 == 0):
                import warnings
                warnings.warn("ravqa_amd: this index cannot run the gpu-fp16 numerics mode (it needs fp16-representable centroids "
                              "and K % 64 == 0); using the CPU-path arithmetic (numerics='cpu') instead", RuntimeWarning)
                self.numerics = "cpu"
        self.max_batch = int(max_batch)
        self._searcher = None       # slot 0: the only one single-chunk calls, the phased protocol and taps use
        self._searcher_key = None
        self._searcher_epoch = 0
        self._side = []             # slots 1..streams-1: (searcher handle, torch stream), same bounds as slot 0
        self._nstreams = max(1, int(streams))
        self._tap_from = None       # the searcher that ran the last chunk
        self._profiled = []         # the searchers of the last profiled search_batch call
        if isinstance(self.arrays, IndexArrays):
            self.codec = _Codec(self.arrays)
            self.embeddings = _Embeddings(self.arrays)
            self.doclens = torch.from_numpy(self.arrays.doclens)
            self.ivf = _Strided(torch.from_numpy(self.arrays.ivf), torch.from_numpy(self.arrays.ivf_lengths))
            self.embeddings_strided = _EmbeddingsStrided(self.embeddings, self.doclens)

    def clone(self, max_batch=None):
        """Another scorer on the SAME resident index with its own native searcher (workspace): what a second in-flight
        sub-batch of the pipelined sharded protocol runs on (the phases of a batch share workspace state)."""
        return type(self)(device_index=self.device_index, max_batch=max_batch or self.max_batch, numerics=self.numerics)

    # ---- native searcher (workspace) management ---------------------------------------------------------
    def _get_searcher(self, nqueries, nq, p):
        key = (max(nqueries, 1), nq, p.ncells, p.ndocs, p.nq_cand)
        if self._searcher is not None and self._searcher_epoch != _native.options_epoch:
            self.close_searcher()  # a switch changed (flmr_set_option): the native searcher snapshots them at creation
        cur = self._searcher_key
        if cur is None or key[0] > cur[0] or key[1] > cur[1] or key[2] > cur[2] or key[3] > cur[3] or key[4] > cur[4]:
            grown = key if cur is None else tuple(max(a, b) for a, b in zip(key, cur))
            self.close_searcher()
            h = C.c_void_p()
            mp = _params(1, grown[2], 0.0, grown[3], grown[4])
            _native.check(self._lib.flmr_searcher_create(self.device_index.handle, grown[0], grown[1], C.byref(mp), C.byref(h)))
            self._searcher, self._searcher_key, self._searcher_epoch = h, grown, _native.options_epoch
            _native.check(self._lib.flmr_searcher_set_numerics(h, self.NUMERICS[self.numerics]))
        return self._searcher

    def _side_slots(self, count):
        """Searchers for slots 1..count (created on first use with slot 0's bounds; close_searcher drops them)."""
        g = self._searcher_key
        while len(self._side) < count:
            h = C.c_void_p()
            mp = _params(1, g[2], 0.0, g[3], g[4])
            _native.check(self._lib.flmr_searcher_create(self.device_index.handle, g[0], g[1], C.byref(mp), C.byref(h)))
            _native.check(self._lib.flmr_searcher_set_numerics(h, self.NUMERICS[self.numerics]))
            self._side.append((h, torch.cuda.Stream()))
        return self._side[:count]

    def close_searcher(self):
        for h, _ in self._side:
            self._lib.flmr_searcher_destroy(h)
        self._side, self._tap_from, self._profiled = [], None, []
        if self._searcher is not None:
            self._lib.flmr_searcher_destroy(self._searcher)
            self._searcher, self._searcher_key = None, None

    def check(self):
        """Wait for the last batch and raise FlmrNativeError if it overflowed the candidate bound or was handed q_lens
        outside [0, nq] (flmr_searcher_check); without this call the error surfaces on the next batch.  EVERY searcher is
        polled (and its flags cleared) before the first error is raised: with streams > 1 a side searcher's stale flag would
        otherwise fail a later batch."""
        first = None
        for h in [self._searcher] + [h for h, _ in self._side]:
            if h is None:
                continue
            try:
                _native.check(self._lib.flmr_searcher_check(h))
            except _native.FlmrNativeError as e:
                first = first or e
        if first is not None:
            raise first

    def supports_query_split(self, Q, k, ncells, thr, ndocs, nq_cand=32):
        """True iff the query-split stage 0 (probe / phase1_probed) runs for this batch shape; depends only on
        replicated data, so every rank of a sharded job gets the same answer (flmr_searcher_probe_supported)."""
        _, _, _, nq, p, s = self._phase_args(Q, k, ncells, thr, ndocs, nq_cand, None)
        ok = C.c_int32(0)
        _native.check(self._lib.flmr_searcher_probe_supported(s, nq, C.byref(p), C.byref(ok)))
        return bool(ok.value)

    def workspace_bytes(self):
        total = 0
        for h in [self._searcher] + [h for h, _ in self._side]:
            if h is None:
                continue
            b = C.c_int64(0)
            _native.check(self._lib.flmr_searcher_workspace_bytes(h, C.byref(b)))
            total += b.value
        return total

    def __del__(self):
        try:
            self.close_searcher()
        except Exception:
            pass

    # ---- batched fast path ---------------------------------------------------------------------------------
    def search_batch(self, Q, k, ncells, centroid_score_threshold, ndocs, nq_cand=32, q_lens=None, profile=False, full_table=False,
                     _after_chunk=None):
        """Q: float32 [n, Nq, 128] (CPU or CUDA).  Returns CUDA tensors (pids i32 [n,k], scores f32 [n,k], counts i32 [n])."""
        if Q.dim() != 3 or Q.size(-1) != self.arrays.dim:
            raise ValueError(f"Q must be [n, Nq, {self.arrays.dim}], got {tuple(Q.shape)}")
        Qd = Q.to(device="cuda", dtype=torch.float32).contiguous()
        n, nq = Qd.size(0), Qd.size(1)
        p = _params(k, ncells, centroid_score_threshold, ndocs, nq_cand)
        s = self._get_searcher(min(n, self.max_batch), nq, p)
        ql = None if q_lens is None else torch.as_tensor(q_lens).to(device="cuda", dtype=torch.int32).contiguous()
        out_p = torch.empty((n, k), dtype=torch.int32, device="cuda")
        out_s = torch.empty((n, k), dtype=torch.float32, device="cuda")
        out_c = torch.empty((n,), dtype=torch.int32, device="cuda")
        B = min(self._searcher_key[0], self.max_batch)
        nchunks = (n + B - 1) // B
        cur = torch.cuda.current_stream()
        slots = [(s, cur)]                          # slot 0 stays on the caller's stream
        if nchunks > 1 and self._nstreams > 1:
            slots += self._side_slots(min(nchunks, self._nstreams) - 1)
        full_table = bool(full_table)
        for h, _ in slots:
            self._lib.flmr_searcher_set_profiling(h, 1 if profile else 0)
            self._lib.flmr_searcher_set_full_table(h, 1 if full_table else 0)  # needed for the CENTROID_SCORES tap
        self.full_table_state = bool(full_table)
        if len(slots) > 1:
            start = cur.record_event()              # inputs / outputs were produced on the caller's stream
            for _, st in slots[1:]:
                st.wait_event(start)
        for i, b0 in enumerate(range(0, n, B)):
            b1 = min(n, b0 + B)
            h, st = slots[i % len(slots)]
            _native.check(self._lib.flmr_search_batch(
                h, C.c_void_p(Qd[b0:b1].data_ptr()), C.c_void_p(ql[b0:b1].data_ptr()) if ql is not None else None,
                b1 - b0, nq, C.byref(p), C.c_void_p(out_p[b0:b1].data_ptr()), C.c_void_p(out_s[b0:b1].data_ptr()),
                C.c_void_p(out_c[b0:b1].data_ptr()), C.c_void_p(st.cuda_stream)))
            self._tap_from = h
            if _after_chunk is not None:
                _after_chunk(i, b0, b1, out_p, out_s, out_c, h, st)
        for _, st in slots[1:]:
            cur.wait_event(st.record_event())
            for t in (Qd, ql, out_p, out_s, out_c):
                if t is not None:
                    t.record_stream(st)             # the caching allocator must not recycle them before the side stream is done
        if profile:   # every searcher profiled since the last stage_ms() read (the library sums per searcher until it is read)
            self._profiled = list({id(h): h for h in self._profiled + [h for h, _ in slots]}.values())
        return out_p, out_s, out_c

    def search_batch_pending(self, Q, k, ncells, centroid_score_threshold, ndocs, nq_cand=32, q_lens=None):
        """search_batch whose results go to pinned HOST memory sub-batch by sub-batch: returns a PendingBatch at once; each
        sub-batch's rows (and the searcher's deferred status words, flmr_searcher_status_async) are copied behind its kernels on
        the launch stream and an event marks them readable, so the host reads sub-batch i while the device computes i+1.  The
        deferred errors of search_batch_checked surface when an affected sub-batch is first read."""
        if self._nstreams > 1:   # sub-batches on several streams finish together: nothing to pipeline
            return PendingBatch.resolved(*self.search_batch_checked(Q, k, ncells, centroid_score_threshold, ndocs, nq_cand, q_lens=q_lens))
        n = Q.size(0)
        hp = torch.empty((n, k), dtype=torch.int32, pin_memory=True)
        hs = torch.empty((n, k), dtype=torch.float32, pin_memory=True)
        hc = torch.empty((n,), dtype=torch.int32, pin_memory=True)
        marks = []

        def after_chunk(j, b0, b1, out_p, out_s, out_c, handle, stream):
            hp[b0:b1].copy_(out_p[b0:b1], non_blocking=True)
            hs[b0:b1].copy_(out_s[b0:b1], non_blocking=True)
            hc[b0:b1].copy_(out_c[b0:b1], non_blocking=True)
            flags = torch.zeros(4, dtype=torch.int32, pin_memory=True)
            _native.check(self._lib.flmr_searcher_status_async(handle, C.c_void_p(flags.data_ptr()), C.c_void_p(stream.cuda_stream)))
            marks.append((b0, b1, stream.record_event(), flags))

        dev = self.search_batch(Q, k, ncells, centroid_score_threshold, ndocs, nq_cand, q_lens=q_lens, _after_chunk=after_chunk)
        return PendingBatch(hp, hs, hc, marks, dev, self.check)

    def search_batch_checked(self, Q, k, ncells, centroid_score_threshold, ndocs, nq_cand=32, q_lens=None):
        """search_batch + check() (a host sync): the deferred device errors of the batch (candidate bound, q_lens range) raise
        here.  Nothing is left to recover on the host: a query with more centroids above the threshold than the searcher keeps
        score rows for (FLMR_ROW_CAP; the reference has no such limit, index_storage.py:116) has its stage 1 recomputed from the
        centroids inside the same batch by the library."""
        out = self.search_batch(Q, k, ncells, centroid_score_threshold, ndocs, nq_cand, q_lens=q_lens)
        self.check()
        return out

    # ---- exact sharded protocol (include/flmr_hip.h: flmr_search_phase1..3) -------------------------------------------
    def _phase_args(self, Q, k, ncells, thr, ndocs, nq_cand, q_lens):
        Qd = Q.to(device="cuda", dtype=torch.float32).contiguous()
        n, nq = Qd.size(0), Qd.size(1)
        p = _params(k, ncells, thr, ndocs, nq_cand)
        if self._searcher_key is not None and self._searcher_key[3] != ndocs:
            self.close_searcher()  # key rows are exactly ndocs wide: the workspace must be created for this ndocs
        s = self._get_searcher(n, nq, p)
        ql = None if q_lens is None else torch.as_tensor(q_lens).to(device="cuda", dtype=torch.int32).contiguous()
        return Qd, ql, n, nq, p, s

    def phase1(self, Q, k, ncells, thr, ndocs, nq_cand=32, q_lens=None):
        """-> int64 tensor [n, ndocs]: this shard's top-ndocs stage-1 keys (score bits << 32 | global pid, 0 = empty)."""
        Qd, ql, n, nq, p, s = self._phase_args(Q, k, ncells, thr, ndocs, nq_cand, q_lens)
        self._phase_state = (Qd, ql, n, nq, p)
        out = torch.empty((n, ndocs), dtype=torch.int64, device="cuda")
        _native.check(self._lib.flmr_search_phase1(s, C.c_void_p(Qd.data_ptr()), C.c_void_p(ql.data_ptr()) if ql is not None else None,
                                                   n, nq, C.byref(p), C.c_void_p(out.data_ptr()), _native.stream_ptr()))
        return out

    def probe_dims(self, Q, k, ncells, thr, ndocs, nq_cand=32):
        """(idx_words, max_cells) of the probe buffers for this batch shape (creates the searcher)."""
        _, _, _, _, _, s = self._phase_args(Q, k, ncells, thr, ndocs, nq_cand, None)
        iw, mc = C.c_int32(0), C.c_int32(0)
        _native.check(self._lib.flmr_searcher_probe_dims(s, C.byref(iw), C.byref(mc)))
        return iw.value, mc.value

    def probe(self, Q, k, ncells, thr, ndocs, q_begin, q_count, nq_cand=32, q_lens=None, out=None):
        """Stage 0 for the query slice [q_begin, q_begin+q_count) of the batch -> (idx_bits i32 [q_count, idx_words],
        cells i32 [q_count, max_cells], ncell i32 [q_count]) on the device (include/flmr_hip.h: flmr_search_probe)."""
        Qd, ql, n, nq, p, s = self._phase_args(Q, k, ncells, thr, ndocs, nq_cand, q_lens)
        iw, mc = C.c_int32(0), C.c_int32(0)
        _native.check(self._lib.flmr_searcher_probe_dims(s, C.byref(iw), C.byref(mc)))
        if out is None:
            out = (torch.empty((q_count, iw.value), dtype=torch.int32, device="cuda"),
                   torch.empty((q_count, mc.value), dtype=torch.int32, device="cuda"),
                   torch.empty((q_count,), dtype=torch.int32, device="cuda"))
        bits, cells, ncell = out
        _native.check(self._lib.flmr_search_probe(s, C.c_void_p(Qd.data_ptr()), C.c_void_p(ql.data_ptr()) if ql is not None else None,
                                                  n, nq, C.byref(p), q_begin, q_count, C.c_void_p(bits.data_ptr()),
                                                  C.c_void_p(cells.data_ptr()), C.c_void_p(ncell.data_ptr()), _native.stream_ptr()))
        return bits, cells, ncell

    def phase1_probed(self, Q, k, ncells, thr, ndocs, idx_bits, cells, ncell, nq_cand=32, q_lens=None):
        """phase1 continuing from the gathered probe state of ALL queries (flmr_search_phase1_probed)."""
        Qd, ql, n, nq, p, s = self._phase_args(Q, k, ncells, thr, ndocs, nq_cand, q_lens)
        self._phase_state = (Qd, ql, n, nq, p)
        bits, cells, ncell = (t.to(device="cuda", dtype=torch.int32).contiguous() for t in (idx_bits, cells, ncell))
        assert bits.size(0) >= n and cells.size(0) >= n and ncell.numel() >= n
        out = torch.empty((n, ndocs), dtype=torch.int64, device="cuda")
        _native.check(self._lib.flmr_search_phase1_probed(s, C.c_void_p(Qd.data_ptr()), C.c_void_p(ql.data_ptr()) if ql is not None else None,
                                                          n, nq, C.byref(p), C.c_void_p(bits.data_ptr()), C.c_void_p(cells.data_ptr()),
                                                          C.c_void_p(ncell.data_ptr()), C.c_void_p(out.data_ptr()), _native.stream_ptr()))
        return out

    def phase2(self, global_s1):
        Qd, ql, n, nq, p = self._phase_state
        g = global_s1.to(device="cuda", dtype=torch.int64).contiguous()
        out = torch.empty((n, p.ndocs), dtype=torch.int64, device="cuda")
        _native.check(self._lib.flmr_search_phase2(self._searcher, C.c_void_p(Qd.data_ptr()), C.c_void_p(ql.data_ptr()) if ql is not None else None,
                                                   n, nq, C.byref(p), C.c_void_p(g.data_ptr()), g.size(1), C.c_void_p(out.data_ptr()),
                                                   _native.stream_ptr()))
        return out

    def phase3(self, global_s2):
        Qd, ql, n, nq, p = self._phase_state
        g = global_s2.to(device="cuda", dtype=torch.int64).contiguous()
        out = torch.empty((n, p.ndocs // 4), dtype=torch.int64, device="cuda")
        _native.check(self._lib.flmr_search_phase3(self._searcher, C.c_void_p(Qd.data_ptr()), C.c_void_p(ql.data_ptr()) if ql is not None else None,
                                                   n, nq, C.byref(p), C.c_void_p(g.data_ptr()), g.size(1), C.c_void_p(out.data_ptr()),
                                                   _native.stream_ptr()))
        return out

    def stage_ms(self):
        """Per-stage HIP-event milliseconds ACCUMULATED over every search_batch(profile=True) call since the previous read
        (flmr_searcher_stage_ms sums per searcher and clears on read), over all sub-batches and all the searchers those
        calls used (with streams > 1 the sub-batches overlap, so the stages add up to more than the wall time)."""
        tot = [0.0] * _native.NUM_STAGES
        for h in (self._profiled or [self._searcher]):
            ms = (C.c_float * _native.NUM_STAGES)()
            _native.check(self._lib.flmr_searcher_stage_ms(h, ms))
            tot = [a + float(b) for a, b in zip(tot, ms)]
        self._profiled = []
        return {self._lib.flmr_stage_name(i).decode(): tot[i] for i in range(_native.NUM_STAGES)}

    def tap(self, what, query=0):
        """Stage output of the last search_batch chunk for `query` (index inside that chunk), as numpy."""
        K = self.arrays.num_centroids
        cap = {_native.TAP_CENTROID_SCORES: K * 128, _native.TAP_IDX_BITS: (K + 31) // 32,
               _native.TAP_CELLS: 1024, _native.TAP_CANDIDATES: self.arrays.num_passages,
               _native.TAP_STAGE1: 8192, _native.TAP_STAGE2: 2048, _native.TAP_DOC_SCORES: 2048,
               _native.TAP_Q_ERR: 32, _native.TAP_Q_ERR_SUM: 1, _native.TAP_STAGE1_FORM: 1}[what]
        dt = {_native.TAP_CENTROID_SCORES: np.float32, _native.TAP_IDX_BITS: np.uint32, _native.TAP_DOC_SCORES: np.float32,
              _native.TAP_Q_ERR: np.float32, _native.TAP_Q_ERR_SUM: np.float32}.get(what, np.int32)
        buf = np.empty(max(cap, 1), dtype=dt)
        cnt = C.c_int64(0)
        _native.check(self._lib.flmr_searcher_tap(self._tap_from or self._searcher, what, query, buf.ctypes.data, cap, C.byref(cnt)))
        out = buf[:cnt.value].copy()
        if what == _native.TAP_CENTROID_SCORES:
            out = out.reshape(K, -1)
        return out

    # ---- reference-shaped API (index_storage.py:67-182) -----------------------------------------------------
    def retrieve(self, config, Q):
        """-> (candidate pids int32 ascending, centroid_scores f32 [K, nq_cand]) for ONE query (index_storage.py:67-80)."""
        Q = Q if Q.dim() == 3 else Q.unsqueeze(0)
        nqc = min(config.query_maxlen, Q.size(1))
        self.search_batch(Q[:1], 1, config.ncells, config.centroid_score_threshold, max(config.ndocs, 4), config.query_maxlen,
                          full_table=True)
        pids = torch.from_numpy(self.tap(_native.TAP_CANDIDATES))
        cs = torch.from_numpy(self.tap(_native.TAP_CENTROID_SCORES)[:, :nqc].copy())
        return pids, cs

    def rank(self, config, Q, filter_fn=None):
        """-> (pids list, scores list), at most ndocs//4 entries, descending score (index_storage.py:86-98)."""
        with torch.inference_mode():
            Q = Q if Q.dim() == 3 else Q.unsqueeze(0)
            if filter_fn is None:
                kk = max(config.ndocs // 4, 1)
                p, s, c = self.search_batch(Q[:1], kk, config.ncells, config.centroid_score_threshold, config.ndocs,
                                            config.query_maxlen)
                n = int(c[0])
                self.check()
                return p[0, :n].tolist(), s[0, :n].tolist()
            # per-query staging so the callable sees the same ascending int32 pid tensor (index_storage.py:90-91)
            pids, centroid_scores = self.retrieve(config, Q)
            pids = filter_fn(pids)
            scores, pids = self.score_pids(config, Q, pids, centroid_scores)
            order = torch.argsort(scores, descending=True, stable=True)
            return pids[order].tolist(), scores[order].tolist()

    def score_pids(self, config, Q, pids, centroid_scores):
        """Pruning (S1+S2) + exact scoring of an explicit candidate list (index_storage.py:100-182)."""
        from . import ops
        pids = torch.as_tensor(pids).to(torch.int32)
        idx = centroid_scores.max(-1).values >= config.centroid_score_threshold
        offsets = self.embeddings_strided.codes_strided.offsets
        fin = ops.filter_pids(pids, centroid_scores, self.embeddings.codes, self.doclens, offsets, idx, config.ndocs,
                              _codes_dev=self._dev("codes"), _offsets_dev=self._dev("offsets"))
        Qd = Q.to("cuda", torch.float32).reshape(-1, self.arrays.dim).contiguous()
        fd = fin.to("cuda")
        out = torch.empty(fd.numel(), dtype=torch.float32, device="cuda")
        if fd.numel():
            _native.check(self._lib.flmr_score_pids(self.device_index.handle, C.c_void_p(Qd.data_ptr()), Qd.size(0),
                                                    C.c_void_p(fd.data_ptr()), fd.numel(), C.c_void_p(out.data_ptr()),
                                                    _native.stream_ptr()))
        return out.cpu(), fin

    # ---- GPU-branch helpers of the reference (index_storage.py:61-65 -> residual_embeddings_strided.py:23-41) ----------
    def _decompress_rows(self, seg_ids, seg_lengths, seg_offsets):
        """flmr_decompress_residuals over explicit (id, length, token offset) segments, then the row normalisation of
        ResidualCodec.decompress (residual.py:268-270: F.normalize(fp32), eps 1e-12) -- fp32 rows on the device."""
        from . import ops
        c = self.codec
        D = ops.decompress_residuals(seg_ids, seg_lengths, seg_offsets, c.bucket_weights, c.reversed_bit_map,
                                     c.decompression_lookup_table, self._dev("residuals"), self._dev("codes"),
                                     self._dev("centroids"), c.dim, c.nbits, _on_device=True)
        return torch.nn.functional.normalize(D, p=2, dim=-1)

    def lookup_pids(self, passage_ids, out_device="cuda", return_mask=False):
        """-> (embeddings_packed f32 [sum doclens, dim] normalised, doclens i64 [n]) for LOCAL passage ids, in the given
        order (index_storage.py:64-65; ResidualEmbeddingsStrided.lookup_pids)."""
        pids = torch.as_tensor(passage_ids).reshape(-1).to(torch.int32)
        D = self._decompress_rows(pids, self.doclens, self.embeddings_strided.codes_strided.offsets)
        return D.to(out_device), self.doclens[pids.long().cpu()]

    def lookup_eids(self, embedding_ids, codes=None, out_device="cuda"):
        """-> normalised embeddings f32 [n, dim] of explicit token ids (index_storage.py:61-62;
        ResidualEmbeddingsStrided.lookup_eids).  `codes` overrides the tokens' centroid ids like the reference's argument."""
        eids = torch.as_tensor(embedding_ids).reshape(-1).long().cpu()
        n = eids.numel()
        if codes is not None:
            # the reference adds the residual of token e to centroid codes[i]: decompress with an index view whose code
            # column is the override (n rows only)
            from . import ops
            c = self.codec
            res = self.embeddings.residuals[eids]
            D = ops.decompress_residuals(torch.arange(n, dtype=torch.int32), torch.ones(n, dtype=torch.int64),
                                         torch.arange(n, dtype=torch.int64), c.bucket_weights, c.reversed_bit_map,
                                         c.decompression_lookup_table, res, torch.as_tensor(codes).reshape(-1).to(torch.int32),
                                         self._dev("centroids"), c.dim, c.nbits, _on_device=True)
            return torch.nn.functional.normalize(D, p=2, dim=-1).to(out_device)
        D = self._decompress_rows(torch.arange(n, dtype=torch.int32), torch.ones(n, dtype=torch.int64), eids)
        return D.to(out_device)

    def _dev(self, name):
        cache = self.__dict__.setdefault("_dev_cache", {})
        if name not in cache:
            src = {"codes": self.embeddings.codes, "offsets": self.embeddings_strided.codes_strided.offsets,
                   "residuals": self.embeddings.residuals, "centroids": self.codec.centroids}[name]
            cache[name] = src.to("cuda")
        return cache[name]
