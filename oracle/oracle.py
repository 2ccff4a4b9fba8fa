"""numpy/ctypes front-end of the CPU oracle (oracle/flmr_oracle.c) and of oracle/_ref.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.

Every function cites the reference file:line it restates ("TPC/" =
third_party/ColBERT/colbert/ under the reference root).
"""
import ctypes as C
import importlib.util
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile liboracle.so (gcc, a second or two).  Safe to call repeatedly."""
    so = os.path.join(HERE, "liboracle.so")
    src = os.path.join(HERE, "flmr_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_candidates.restype = C.c_int64
        _LIB.orc_decompress_residuals.restype = C.c_int64
        _LIB.orc_segmented_lookup.restype = C.c_int64
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# ---------------------------------------------------------------------------------------------
# codec tables (TPC/indexing/codecs/residual.py:51-95)
# ---------------------------------------------------------------------------------------------
def codec_tables(nbits):
    """reversed_bit_map u8[256] and decompression_lookup_table u8[256, 8/nbits].

    reversed_bit_map[i]: each nbits-wide group of byte i has its bits reversed, groups keep their place
    (residual.py:51-73).  decompression_lookup_table[x] = base-2^nbits digits of x, most significant
    first (itertools.product order, residual.py:77-89)."""
    vpb = 8 // nbits
    rev = np.zeros(256, dtype=np.uint8)
    lut = np.zeros((256, vpb), dtype=np.uint8)
    mask = (1 << nbits) - 1
    for i in range(256):
        z = 0
        for g in range(vpb):  # group g counted from the most significant end
            x = (i >> (8 - nbits * (g + 1))) & mask
            y = 0
            for b in range(nbits):
                y |= ((x >> b) & 1) << (nbits - 1 - b)
            z |= y << (8 - nbits * (g + 1))
            lut[i, g] = (i >> (8 - nbits * (g + 1))) & mask
        rev[i] = z
    return rev, lut


class OracleIndex:
    """In-memory index in the reference's CPU layout: codes i32[N], residuals
    u8[N,B], doclens i64, offsets i64[+1], ivf i32 + ivf_offsets i64[K+1], centroids f32, tables."""

    class _S(C.Structure):
        _fields_ = [("dim", C.c_int32), ("nbits", C.c_int32), ("K", C.c_int32), ("num_passages", C.c_int64),
                    ("codes", C.c_void_p), ("residuals", C.c_void_p), ("doclens", C.c_void_p),
                    ("offsets", C.c_void_p), ("ivf", C.c_void_p), ("ivf_offsets", C.c_void_p),
                    ("centroids", C.c_void_p), ("bucket_weights", C.c_void_p),
                    ("reversed_bit_map", C.c_void_p), ("lut", C.c_void_p)]

    def __init__(self, dim, nbits, codes, residuals, doclens, ivf, ivf_lengths, centroids, bucket_weights):
        self.dim, self.nbits = int(dim), int(nbits)
        self.codes = _c(codes, np.int32)
        self.residuals = _c(residuals, np.uint8)
        self.doclens = _c(doclens, np.int64)
        self.offsets = np.concatenate([[0], np.cumsum(self.doclens)]).astype(np.int64)
        self.ivf = _c(ivf, np.int32)
        self.ivf_lengths = _c(ivf_lengths, np.int64)
        self.ivf_offsets = np.concatenate([[0], np.cumsum(self.ivf_lengths)]).astype(np.int64)
        self.centroids = _c(centroids, np.float32)  # fp16 values widened (residual.py:29)
        self.K = self.centroids.shape[0]
        self.bucket_weights = _c(bucket_weights, np.float32)
        self.reversed_bit_map, self.lut = codec_tables(self.nbits)
        self.num_passages = len(self.doclens)
        self._s = self._S(self.dim, self.nbits, self.K, self.num_passages, _p(self.codes), _p(self.residuals),
                          _p(self.doclens), _p(self.offsets), _p(self.ivf), _p(self.ivf_offsets),
                          _p(self.centroids), _p(self.bucket_weights), _p(self.reversed_bit_map), _p(self.lut))

    @classmethod
    def from_golden(cls, z):
        return cls(int(z["meta.dim"]), int(z["meta.nbits"]), z["index.codes"], z["index.residuals"],
                   z["index.doclens"], z["index.ivf"], z["index.ivf_lengths"],
                   z["index.centroids_f16"].astype(np.float32), z["index.bucket_weights"])

    # ---- stages -------------------------------------------------------------------------
    def centroid_scores(self, Qc):
        Qc = _c(Qc, np.float32)
        out = np.empty((self.K, Qc.shape[0]), dtype=np.float32)
        lib().orc_centroid_scores(_p(self.centroids), _p(Qc), self.K, Qc.shape[0], self.dim, _p(out))
        return out

    def candidates(self, cells):
        cells = _c(cells, np.int32)
        out = np.empty(self.num_passages, dtype=np.int32)
        P = lib().orc_candidates(_p(cells), len(cells), _p(self.ivf), _p(self.ivf_offsets),
                                 C.c_int64(self.num_passages), _p(out))
        return out[:P].copy()

    def filter_pass(self, pids, cs, idx, n_keep):
        pids, cs = _c(pids, np.int32), _c(cs, np.float32)
        idx_ = None if idx is None else _c(idx, np.uint8)
        op = np.empty(max(n_keep, 1), dtype=np.int32)
        os_ = np.empty(max(n_keep, 1), dtype=np.float32)
        n = lib().orc_filter_pass(_p(pids), C.c_int64(len(pids)), _p(cs), cs.shape[1], _p(self.codes),
                                  _p(self.doclens), _p(self.offsets), _p(idx_), n_keep, _p(op), _p(os_))
        return op[:n].copy(), os_[:n].copy()

    def filter_pids(self, pids, cs, idx, ndocs):
        pids, cs, idx = _c(pids, np.int32), _c(cs, np.float32), _c(idx, np.uint8)
        op = np.empty(max(ndocs // 4, 1), dtype=np.int32)
        n = lib().orc_filter_pids(_p(pids), C.c_int64(len(pids)), _p(cs), cs.shape[1], _p(self.codes),
                                  _p(self.doclens), _p(self.offsets), _p(idx), ndocs, _p(op))
        return op[:n].copy()

    def decompress(self, pids):
        pids = _c(pids, np.int32)
        n = int(self.doclens[pids].sum()) if len(pids) else 0
        out = np.empty((n, self.dim), dtype=np.float32)
        w = lib().orc_decompress_residuals(_p(pids), len(pids), _p(self.doclens), _p(self.offsets),
                                           _p(self.bucket_weights), _p(self.reversed_bit_map), _p(self.lut),
                                           _p(self.residuals), _p(self.codes), _p(self.centroids), self.dim,
                                           self.nbits, _p(out))
        assert w == n
        return out

    def rank(self, Q, ncells, thr, ndocs, nq_cand=32):
        Q = _c(Q, np.float32)
        cap = max(ndocs // 4, 1)
        op, os_ = np.empty(cap, dtype=np.int32), np.empty(cap, dtype=np.float32)
        nc = C.c_int64(0)
        n = lib().orc_rank(C.byref(self._s), _p(Q), Q.shape[0], nq_cand, ncells, C.c_float(thr), ndocs, _p(op),
                           _p(os_), C.byref(nc))
        return op[:n].copy(), os_[:n].copy(), int(nc.value)

    def search_batch(self, Q, k, ncells, thr, ndocs, nq_cand=32, threads=None):
        Q = _c(Q, np.float32)
        nqr, nq = Q.shape[0], Q.shape[1]
        lib().orc_set_threads(int(threads) if threads else effective_cpus())
        op = np.empty((nqr, k), dtype=np.int32)
        os_ = np.empty((nqr, k), dtype=np.float32)
        oc = np.empty(nqr, dtype=np.int32)
        lib().orc_search_batch(C.byref(self._s), _p(Q), nqr, nq, nq_cand, ncells, C.c_float(thr), ndocs, k, _p(op),
                               _p(os_), _p(oc))
        return op, os_, oc


def select_cells(cs, ncells):
    cs = _c(cs, np.float32)
    out = np.empty(cs.shape[1] * ncells, dtype=np.int32)
    n = lib().orc_select_cells(_p(cs), cs.shape[0], cs.shape[1], ncells, _p(out))
    return out[:n].copy()


def idx_mask(cs, thr):
    cs = _c(cs, np.float32)
    out = np.empty(cs.shape[0], dtype=np.uint8)
    lib().orc_idx_mask(_p(cs), cs.shape[0], cs.shape[1], C.c_float(thr), _p(out))
    return out.astype(bool)


def normalize_rows(D):
    D = _c(D, np.float32).copy()
    lib().orc_normalize_rows(_p(D), C.c_int64(D.shape[0]), D.shape[1])
    return D


def segmented_maxsim(scores, lengths):
    scores, lengths = _c(scores, np.float32), _c(lengths, np.int64)
    out = np.empty(len(lengths), dtype=np.float32)
    lib().orc_segmented_maxsim(_p(scores), _p(lengths), len(lengths), scores.shape[1], _p(out))
    return out


def maxsim_packed(D, Q, lengths):
    D, Q, lengths = _c(D, np.float32), _c(Q, np.float32), _c(lengths, np.int64)
    out = np.empty(len(lengths), dtype=np.float32)
    lib().orc_maxsim_packed(_p(D), _p(Q), _p(lengths), len(lengths), Q.shape[0], Q.shape[1], _p(out))
    return out


def colbert_score_padded(Q, D, mask):
    Q, D = _c(Q, np.float32), _c(D, np.float32)
    mask = _c(mask, np.uint8)
    B, Ld, dim = D.shape
    out = np.empty(B, dtype=np.float32)
    lib().orc_colbert_score_padded(_p(Q), Q.shape[0], Q.shape[1], _p(D), _p(mask), B, Ld, dim, _p(out))
    return out


def segmented_lookup(inp, lengths, offsets):
    inp = np.ascontiguousarray(inp)
    lengths, offsets = _c(lengths, np.int64), _c(offsets, np.int64)
    row_bytes = inp.dtype.itemsize * (int(np.prod(inp.shape[1:])) if inp.ndim > 1 else 1)
    n = int(lengths.sum())
    out = np.empty((n,) + inp.shape[1:], dtype=inp.dtype)
    lib().orc_segmented_lookup(_p(inp), C.c_int64(row_bytes), _p(lengths), _p(offsets), len(lengths), _p(out))
    return out


# ---------------------------------------------------------------------------------------------
# oracle/_ref: the reference's own C++ ops (compiled in place by build_ref.py) + the torch-CPU glue
# of IndexScorer.rank.  Used to cross-check the restatement and as bench.py's cpu_baseline
# ("reference" kind).  Needs torch (CPU) but never /root/reference at run time.
# ---------------------------------------------------------------------------------------------
_REF = {}


def effective_cpus():
    """CPUs this process may actually use: min(affinity, cgroup CPU quota).  The GPU boxes show 256 hardware threads under a
    16-CPU quota: a pool sized from nproc spends its time throttled (a 256-thread OpenMP loop measured no faster than 8)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def ref_available():
    return all(os.path.exists(os.path.join(HERE, "_ref", n + ".so")) for n in
               ("filter_pids_cpp", "decompress_residuals_cpp", "segmented_lookup_cpp", "segmented_maxsim_cpp"))


def ref_op(name):
    """Return the pybind function `name` (e.g. 'filter_pids_cpp') from oracle/_ref/<name>.so."""
    if name not in _REF:
        import torch  # noqa: F401  (the extension links libtorch)
        path = os.path.join(HERE, "_ref", name + ".so")
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _REF[name] = getattr(mod, name)
    return _REF[name]


class RefCpuScorer:
    """IndexScorer.rank with the reference's compiled C++ stages and torch-CPU ops for the Python glue
    (TPC/search/index_storage.py:86-182, candidate_generation.py:12-64, colbert.py:289-311)."""

    # The reference ops spawn at::get_num_threads() pthreads per call (filter_pids.cpp:85-104); on a 256-thread host the
    # default is ~9x slower than 8 threads (bench.py cpu_baseline.value_all_threads), and results are bit-identical for any
    # thread count, so the checker pins itself to 8 unless told otherwise.
    THREADS = int(os.environ.get("FLMR_REF_THREADS", "8"))

    def __init__(self, oi: OracleIndex, threads=None):
        import torch
        self.t = torch
        threads = self.THREADS if threads is None else threads
        if threads > 0 and torch.get_num_threads() != threads:
            torch.set_num_threads(threads)
        self.codes = torch.from_numpy(oi.codes)
        self.residuals = torch.from_numpy(oi.residuals)
        self.doclens = torch.from_numpy(oi.doclens)
        self.offsets = torch.from_numpy(oi.offsets)
        self.ivf = torch.from_numpy(oi.ivf)
        self.ivf_lengths = torch.from_numpy(oi.ivf_lengths)
        self.ivf_offsets = torch.from_numpy(oi.ivf_offsets)
        self.centroids = torch.from_numpy(oi.centroids)
        self.bucket_weights = torch.from_numpy(oi.bucket_weights)
        self.rbm = torch.from_numpy(oi.reversed_bit_map)
        self.lut = torch.from_numpy(oi.lut)
        self.dim, self.nbits = oi.dim, oi.nbits
        self.filter_pids = ref_op("filter_pids_cpp")
        self.decompress = ref_op("decompress_residuals_cpp")
        self.lookup = ref_op("segmented_lookup_cpp")
        self.maxsim = ref_op("segmented_maxsim_cpp")

    def rank(self, Q, ncells, thr, ndocs, nq_cand=32):
        torch = self.t
        with torch.inference_mode():
            Qc = Q[:nq_cand]
            scores = self.centroids @ Qc.T
            if ncells == 1:
                cells = scores.argmax(dim=0, keepdim=True).permute(1, 0)
            else:
                cells = scores.topk(ncells, dim=0, sorted=False).indices.permute(1, 0)
            cells = cells.flatten().contiguous().unique(sorted=False).long()
            pids = self.lookup(self.ivf, cells, self.ivf_lengths[cells], self.ivf_offsets[cells])
            pids = torch.unique_consecutive(pids.sort().values)
            idx = scores.max(-1).values >= thr
            fin = self.filter_pids(pids, scores, self.codes, self.doclens, self.offsets, idx, ndocs)
            D = self.decompress(fin, self.doclens, self.offsets, self.bucket_weights, self.rbm, self.lut,
                                self.residuals, self.codes, self.centroids, self.dim, self.nbits)
            D = torch.nn.functional.normalize(D.to(torch.float32), p=2, dim=-1)
            sc = self.maxsim(D @ Q.T, self.doclens[fin.long()])
            srt = sc.sort(descending=True)
            return fin[srt.indices].tolist(), srt.values.tolist(), int(pids.numel())


# ---- the reference's CUDA-branch arithmetic (FLMR_NUMERICS_GPU_FP16), restated with numpy float16 ---------------
def f16(x):
    """fp32 -> nearest fp16 (ties to even, overflow -> +-inf) -> fp32: the value a half tensor holds."""
    with np.errstate(over="ignore"):
        return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


class GpuNumericsOracle:
    """CPU restatement of IndexScorer.rank with use_gpu=True (TPC/search/index_storage.py:86-98,113-158,176-177,
    candidate_generation.py:12-64, indexing/codecs/residual.py:242-278 + decompress_residuals.cu, modeling/colbert.py:235-263,
    289-311): fp16 tensors where that branch holds fp16, fp32 accumulation where the CUDA GEMM / sum accumulates in fp32.
    Where torch leaves the order open (topk / sort among equal scores) ties go to the larger pid, as in the HIP build.
    TEST INFRASTRUCTURE: pinned to the reference's own expressions evaluated on CPU half tensors by
    tests/test_oracle_golden.py::test_gpu_numerics_oracle_vs_reference_expressions (tests/golden/gpu_numerics.npz); parity
    with the reference's CUDA kernels themselves is UNPINNED (no CUDA device in the build environment)."""

    PAD = np.float32(-10000.0)   # half(-9999)

    def __init__(self, oi: OracleIndex):
        self.oi = oi
        self.bw16 = f16(oi.bucket_weights)

    def centroid_scores_raw(self, Q, nq_cand=32):
        return self.oi.centroids @ f16(Q[:nq_cand]).T            # fp32 accumulation of fp16 x fp16 products, [K, nqc]

    def cells(self, raw, ncells):
        """per column the ncells best centroids by (value desc, index asc) of the UNROUNDED products (a valid choice among
        the fp16 ties torch.topk resolves arbitrarily), ascending unique."""
        out = set()
        for k in range(raw.shape[1]):
            order = np.lexsort((np.arange(raw.shape[0]), -raw[:, k]))
            out.update(order[:ncells].tolist())
        return np.array(sorted(out), dtype=np.int32)

    def idx(self, cs16, thr):
        return cs16.max(-1) >= f16(np.float32(thr))

    def approx_scores(self, cs16, pids, idx=None):
        """per passage: sum over columns (fp32 accumulation, fp16 result) of the per-column maximum over its codes
        (restricted to idx when given), -9999 -> half -10000 where no code qualifies."""
        oi, nqc = self.oi, cs16.shape[1]
        out = np.empty(len(pids), dtype=np.float32)
        for j, p in enumerate(np.asarray(pids).tolist()):
            c = oi.codes[oi.offsets[p]:oi.offsets[p + 1]]
            if idx is not None:
                c = c[idx[c]]
            col = cs16[c].max(0) if len(c) else np.full(nqc, self.PAD, dtype=np.float32)
            col = np.maximum(col, self.PAD)
            s = np.float32(0.0)
            for k in range(nqc):
                s = np.float32(s + col[k])
            out[j] = f16(s)
        return out

    @staticmethod
    def top(scores, pids, n):
        """the n best by (score, pid) descending, in that order"""
        pids = np.asarray(pids)
        order = np.lexsort((-pids.astype(np.int64), -scores.astype(np.float64)))
        return pids[order[:n]], scores[order[:n]]

    def embeddings(self, pids):
        oi = self.oi
        vpb = 8 // oi.nbits
        rows = np.concatenate([np.arange(oi.offsets[p], oi.offsets[p + 1]) for p in np.asarray(pids).tolist()] or [np.zeros(0, np.int64)]).astype(np.int64)
        lut = oi.lut.reshape(256, vpb)
        w = self.bw16[lut[oi.reversed_bit_map[oi.residuals[rows]]].reshape(len(rows), -1)]
        D = f16(w + oi.centroids[oi.codes[rows]])                            # half(weight) + half(centroid) in half
        nrm = f16(np.sqrt((D.astype(np.float32) ** 2).sum(-1, dtype=np.float32)))
        with np.errstate(divide="ignore", invalid="ignore"):
            Dn = np.where(nrm[:, None] > 0, f16(D / nrm[:, None]), np.float32(0.0))
        return Dn.astype(np.float32)

    def doc_scores(self, Q, pids):
        oi = self.oi
        Dn = self.embeddings(pids)
        sc = f16(Dn @ f16(Q).T)                                               # [tokens, Nq] half
        out = np.empty(len(pids), dtype=np.float32)
        t = 0
        for j, p in enumerate(np.asarray(pids).tolist()):
            n = int(oi.doclens[p])
            col = sc[t:t + n].max(0) if n else np.full(Q.shape[0], self.PAD, dtype=np.float32)
            t += n
            s = np.float32(0.0)
            for k in range(Q.shape[0]):
                s = np.float32(s + col[k])
            out[j] = f16(s)
        return out

    def rank(self, Q, ncells, thr, ndocs, nq_cand=32):
        Q = np.asarray(Q, dtype=np.float32)
        raw = self.centroid_scores_raw(Q, nq_cand)
        cs16 = f16(raw)
        cand = self.oi.candidates(self.cells(raw, ncells))
        idx = self.idx(cs16, thr)
        s1 = self.approx_scores(cs16, cand, idx)
        p1, _ = self.top(s1, cand, ndocs) if ndocs < len(cand) else (cand, s1)
        s2 = self.approx_scores(cs16, p1)
        p2, _ = self.top(s2, p1, ndocs // 4) if ndocs // 4 < len(p1) else (p1, s2)
        sc = self.doc_scores(Q, p2)
        fp, fs = self.top(sc, p2, len(p2))
        return fp, fs, len(cand)
