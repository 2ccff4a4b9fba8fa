"""torch-tensor front-ends of the op-level C-ABI entry points, with the reference's pybind signatures:

    filter_pids(pids, centroid_scores, codes, doclens, offsets, idx, nfiltered_docs)      filter_pids.cpp:126-164
    decompress_residuals(pids, lengths, offsets, bucket_weights, reversed_bit_map,
                         bucket_weight_combinations, binary_residuals, codes, centroids, dim, nbits)
                                                                                   decompress_residuals.cpp:80-155
    segmented_lookup(input, pids, lengths, offsets)                                segmented_lookup.cpp:127-144
    segmented_maxsim(scores, lengths)                                              segmented_maxsim.cpp:49-93

Inputs may be CPU tensors (as the reference passes them): they are staged to HBM, the HIP kernel runs, and the
result comes back as a CPU tensor like the reference's.  CUDA inputs are used in place.  There is no CPU
implementation behind these functions.
"""
import ctypes as C

import torch

from . import _native


def _d(t, dtype=None):
    t = torch.as_tensor(t)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.to("cuda").contiguous()


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def filter_pids(pids, centroid_scores, codes, doclens, offsets, idx, nfiltered_docs, _codes_dev=None, _offsets_dev=None):
    lib = _native.load()
    pd, cs = _d(pids, torch.int32), _d(centroid_scores, torch.float32)
    cd = _codes_dev if _codes_dev is not None else _d(codes, torch.int32)
    od = _offsets_dev if _offsets_dev is not None else _d(offsets, torch.int64)
    dl = _d(doclens, torch.int64)
    ix = _d(idx, torch.bool).view(torch.uint8)
    out = torch.empty(max(nfiltered_docs // 4, 1), dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    _native.check(lib.flmr_filter_pids(_p(pd), pd.numel(), _p(cs), cs.size(0), cs.size(1), _p(cd), _p(dl), _p(od), _p(ix),
                                       int(nfiltered_docs), _p(out), _p(cnt), _native.stream_ptr()))
    return out[: int(cnt.item())].cpu()


def decompress_residuals(pids, lengths, offsets, bucket_weights, reversed_bit_map, bucket_weight_combinations,
                         binary_residuals, codes, centroids, dim, nbits, _on_device=False):
    lib = _native.load()
    pd = _d(pids, torch.int32)
    ln, of = _d(lengths, torch.int64), _d(offsets, torch.int64)
    nrows = int(torch.as_tensor(lengths).long()[torch.as_tensor(pids).long()].sum()) if pd.numel() else 0
    out = torch.zeros((nrows, dim), dtype=torch.float32, device="cuda")
    # keep every staged tensor referenced until the (synchronous) call returns
    bw, rbm, lut = _d(bucket_weights, torch.float32), _d(reversed_bit_map, torch.uint8), _d(bucket_weight_combinations, torch.uint8)
    res, cod, cen = _d(binary_residuals, torch.uint8), _d(codes, torch.int32), _d(centroids, torch.float32)
    if pd.numel():
        _native.check(lib.flmr_decompress_residuals(
            _p(pd), pd.numel(), _p(ln), _p(of), _p(bw), _p(rbm), _p(lut), _p(res), _p(cod), _p(cen), int(dim), int(nbits),
            _p(out), nrows, None, _native.stream_ptr()))
    return out if _on_device else out.cpu()


def segmented_lookup(input, pids, lengths, offsets):
    lib = _native.load()
    inp = torch.as_tensor(input)
    ind = inp.to("cuda").contiguous()
    ln, of = _d(lengths, torch.int64), _d(offsets, torch.int64)
    nrows = int(torch.as_tensor(lengths).long().sum())
    out = torch.zeros((nrows,) + tuple(inp.shape[1:]), dtype=inp.dtype, device="cuda")
    row_bytes = inp.element_size() * (int(torch.tensor(inp.shape[1:]).prod()) if inp.dim() > 1 else 1)
    if ln.numel():
        _native.check(lib.flmr_segmented_lookup(_p(ind), row_bytes, _p(ln), _p(of), ln.numel(), _p(out), nrows, None,
                                                _native.stream_ptr()))
    return out.cpu() if not inp.is_cuda else out


def segmented_maxsim(scores, lengths):
    lib = _native.load()
    sc, ln = _d(scores, torch.float32), _d(lengths, torch.int64)
    out = torch.zeros(ln.numel(), dtype=torch.float32, device="cuda")
    if ln.numel():
        _native.check(lib.flmr_segmented_maxsim(_p(sc), _p(ln), ln.numel(), sc.size(1), _p(out), _native.stream_ptr()))
    return out.cpu() if not torch.as_tensor(scores).is_cuda else out


def colbert_score_padded(Q, D_padded, D_mask):
    """Forward-only padded MaxSim (colbert.py:268-286): Q [1|B, Nq, d], D [B, Ld, d], mask [B, Ld(,1)] -> [B]."""
    lib = _native.load()
    Qd, Dd = _d(Q, torch.float32), _d(D_padded, torch.float32)
    B, Ld, dim = Dd.shape
    md = _d(torch.as_tensor(D_mask).reshape(B, Ld), torch.bool).view(torch.uint8)
    out = torch.empty(B, dtype=torch.float32, device="cuda")
    _native.check(lib.flmr_colbert_score_padded(_p(Qd), Qd.size(0), Qd.size(1), _p(Dd), _p(md), B, Ld, dim, _p(out),
                                                _native.stream_ptr()))
    return out


def colbert_score_cross(Q, D_padded, D_mask):
    """Every query against every document in ONE launch: Q [nqueries, Nq, d], D [B, Ld, d], mask [B, Ld(,1)] -> [nqueries, B]
    (the rate matrix of the executor's exhaustive search, FLMR_executor.py:799-847)."""
    lib = _native.load()
    Qd, Dd = _d(Q, torch.float32), _d(D_padded, torch.float32)
    B, Ld, dim = Dd.shape
    md = _d(torch.as_tensor(D_mask).reshape(B, Ld), torch.bool).view(torch.uint8)
    out = torch.empty((Qd.size(0), B), dtype=torch.float32, device="cuda")
    _native.check(lib.flmr_colbert_score_cross(_p(Qd), Qd.size(0), Qd.size(1), _p(Dd), _p(md), B, Ld, dim, _p(out),
                                               _native.stream_ptr()))
    return out


def colbert_colmax_padded(Q, D_padded, D_mask):
    """The per-column maxima [B, Nq] of the padded MaxSim (-9999 padding) before their sum: what `colbert_score_reduce`
    reduces -- by a plain sum for 'colbert', by top-k sums for 'flipr' (colbert.py:235-263)."""
    lib = _native.load()
    Qd, Dd = _d(Q, torch.float32), _d(D_padded, torch.float32)
    B, Ld, dim = Dd.shape
    md = _d(torch.as_tensor(D_mask).reshape(B, Ld), torch.bool).view(torch.uint8)
    out = torch.empty((B, Qd.size(1)), dtype=torch.float32, device="cuda")
    _native.check(lib.flmr_colbert_colmax_padded(_p(Qd), Qd.size(0), Qd.size(1), _p(Dd), _p(md), B, Ld, dim, _p(out),
                                                 _native.stream_ptr()))
    return out


def merge_topk(scores, pids):
    """scores f32 / pids i32 [nshards, nqueries, k] (CUDA) -> merged (scores, pids, counts) [nqueries, k]."""
    lib = _native.load()
    sc, pd = _d(scores, torch.float32), _d(pids, torch.int32)
    R, n, k = sc.shape
    os_ = torch.empty((n, k), dtype=torch.float32, device="cuda")
    op = torch.empty((n, k), dtype=torch.int32, device="cuda")
    oc = torch.empty((n,), dtype=torch.int32, device="cuda")
    _native.check(lib.flmr_merge_topk(_p(sc), _p(pd), R, n, k, _p(os_), _p(op), _p(oc), _native.stream_ptr()))
    return os_, op, oc


def topn_keys(keys, n, ordered=True):
    """keys int64 [nq, m] (u64 bit patterns: score bits << 32 | pid, 0 = empty) -> the n largest per row, descending;
    ordered=False: the same set in arbitrary order (radix select, any m)."""
    lib = _native.load()
    kd = keys.to("cuda").contiguous()
    out = torch.empty((kd.size(0), n), dtype=torch.int64, device="cuda")
    if not ordered:
        _native.check(lib.flmr_select_keys(_p(kd), kd.size(0), kd.size(1), int(n), _p(out), _native.stream_ptr()))
        return out
    if kd.size(1) > 2048 and kd.size(1) > 2 * n:
        # long rows: radix-select the n survivors first, then sort only those (a bitonic sort of 8192 keys per row costs
        # five times as much); the result is the same descending list
        kd = topn_keys(kd, n, ordered=False)
    _native.check(lib.flmr_topn_keys(_p(kd), kd.size(0), kd.size(1), int(n), _p(out), None, _native.stream_ptr()))
    return out


def unpack_keys(keys, k):
    """descending keys int64 [nq, n] -> (pids i32 [nq,k], scores f32 [nq,k], counts i32 [nq])."""
    lib = _native.load()
    kd = keys.to("cuda").contiguous()
    nq = kd.size(0)
    op = torch.empty((nq, k), dtype=torch.int32, device="cuda")
    os_ = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    oc = torch.empty((nq,), dtype=torch.int32, device="cuda")
    _native.check(lib.flmr_unpack_keys(_p(kd), nq, kd.size(1), int(k), _p(op), _p(os_), _p(oc), _native.stream_ptr()))
    return op, os_, oc


# ---- index build (include/flmr_hip.h "Index build ops"; residual.py:169-222) -------------------------------------------
def nearest_centroids(embs, centroids):
    """codes int32 [N] = argmax_c centroids[c] . embs[t] (residual.py:206-216); centroids must be fp16-representable."""
    lib = _native.load()
    e, c = _d(embs, torch.float32), _d(centroids, torch.float32)
    assert e.dim() == 2 and e.size(1) == 128 and c.size(1) == 128
    out = torch.empty(e.size(0), dtype=torch.int32, device="cuda")
    _native.check(lib.flmr_nearest_centroids(_p(c), c.size(0), _p(e), e.size(0), _p(out), _native.stream_ptr()))
    return out


def compress_residuals(embs, centroids, codes, bucket_cutoffs, nbits):
    """packed residual bytes uint8 [N, 16*nbits] for the given codes (residual.py:186-204)."""
    lib = _native.load()
    e, c = _d(embs, torch.float32), _d(centroids, torch.float32)
    cd, cut = _d(codes, torch.int32), _d(bucket_cutoffs, torch.float32)
    assert cut.numel() == 2 ** nbits - 1, "bucket_cutoffs must hold 2^nbits - 1 values"
    out = torch.empty((e.size(0), 16 * nbits), dtype=torch.uint8, device="cuda")
    _native.check(lib.flmr_compress_residuals(_p(c), c.size(0), _p(e), _p(cd), e.size(0), _p(cut), int(nbits), _p(out),
                                              _native.stream_ptr()))
    return out


def compress(embs, centroids, bucket_cutoffs, nbits):
    """ResidualCodec.compress on the GPU -> (codes int32 [N], residuals uint8 [N, 16*nbits])."""
    codes = nearest_centroids(embs, centroids)
    return codes, compress_residuals(embs, centroids, codes, bucket_cutoffs, nbits)


def build_ivf(codes, doclens, K):
    """(ivf pids int32 [sum unique], ivf_lengths int64 [K]) of an index on the GPU: optimize_ivf (TPC/indexing/utils.py:8-53)
    through flmr_build_ivf -- a stable device radix sort of (code, pid) by code, run flags, compaction."""
    lib = _native.load()
    cd = _d(codes, torch.int32)
    dl = torch.as_tensor(doclens).to(device="cuda", dtype=torch.int64)
    offsets = torch.zeros(dl.numel() + 1, dtype=torch.int64, device="cuda")
    offsets[1:] = torch.cumsum(dl, 0)
    n = cd.numel()
    assert int(offsets[-1]) == n, "doclens must sum to the number of codes"
    ivf = torch.empty(max(n, 1), dtype=torch.int32, device="cuda")
    lengths = torch.empty(int(K), dtype=torch.int64, device="cuda")
    import ctypes as C
    total = C.c_int64(0)
    _native.check(lib.flmr_build_ivf(_p(cd), n, _p(offsets), dl.numel(), int(K), _p(ivf), _p(lengths), C.byref(total), _native.stream_ptr()))
    return ivf[: total.value].clone(), lengths

