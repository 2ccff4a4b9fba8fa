"""Pins the fp16-split MFMA arithmetic ("f32 results; fp16-split MFMA, fp32 accumulate") against an fp64 ground truth.

Every split kernel -- stage 0 (`s0_centroid_scores_f16`), stage 2 (`filter_stage2_lds_kernel`, recompute), the fused
MaxSim (`maxsim_f16_kernel` / `maxsim_f16_multiq_kernel`) and the padded scorer (`colbert_score_padded_mfma_kernel`) --
computes c.q as mfma16(c, q_hi) + 2^-11 mfma16(c, q_lo) with fp32 accumulation (csrc/flmr_stage0.hip:138-149).  The claim
tested here: on the same inputs its error against the fp64 dot product is NO LARGER than that of a plain k-ascending fp32
loop (the CPU oracle's / the VALU kernel's arithmetic -- one member of the family of "valid fp32 summation orders" the
reference's BLAS belongs to).  Inputs cover unit-norm query rows, FLMR's un-normalised visual-token rows (|q| up to ~16),
rows so small that q_lo is subnormal in fp16, and mixed-magnitude rows.

One caveat is part of the contract and is asserted as such: fp16 has a subnormal quantum of 2^-24, so a query component
below 2^-13 keeps an ABSOLUTE representation error of up to 2^-35 in the split (instead of 2^-22 relative).  For a whole
dot product that is an absolute floor of ~3e-11 -- nine orders below the pruning thresholds (0.4 .. 0.5) and seven below the
1e-4 score tolerance; it only shows for query rows of norm < 1e-3, which neither normalised text tokens nor FLMR's visual
tokens produce.  `FLOOR` below is that term (per score column).

The adversarial-threshold test drives `centroid_score_threshold` to within one ulp of actual table maxima: the pruning
predicate is `max_k cs[c,k] >= thr` (index_storage.py:116, `>=`), an exact function of the table's bits.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

FLOOR = 2e-10   # absolute error floor of the split per 128-d dot product (fp16 subnormal quantum, see module docstring)


@pytest.fixture(scope="module")
def env():
    import torch
    import ravqa_amd
    from ravqa_amd import _native, ops, synth
    from ravqa_amd.scorer import IndexScorer
    assert torch.cuda.is_available()
    _native.load(require_device=True)
    corpus = synth.make_corpus(3000, (20, 90), 1024, 4, seed=21, device="cuda")
    arrays = synth.corpus_to_arrays(corpus)
    return dict(torch=torch, nat=_native, ops=ops, synth=synth, corpus=corpus, arrays=arrays,
                scorer=IndexScorer(arrays=arrays, max_batch=8))


def _query_family(torch, corpus, synth):
    """[5, 32, 128]: unit rows; rows scaled to |q| = 16 (un-normalised visual tokens); tiny rows (q_lo subnormal in
    fp16); per-row mixed magnitudes 1e-3 .. 8; unit rows with a few exactly-fp16 components (q_lo == 0)."""
    Q, _ = synth.make_queries(corpus, 5, 32, seed=8)
    g = torch.Generator(device="cuda").manual_seed(3)
    Q = Q.clone()
    Q[1] *= 16.0
    Q[2] *= 3.0e-5
    Q[3] *= torch.logspace(-3, 0.9, 32, device="cuda").unsqueeze(1)
    Q[4] = Q[4].half().float()
    Q[4, :, ::3] += 1e-4 * torch.randn(32, 43, generator=g, device="cuda")
    return Q.contiguous()


def _errs(got, fp32_loop, truth64):
    e_split = np.abs(got.astype(np.float64) - truth64)
    e_loop = np.abs(fp32_loop.astype(np.float64) - truth64)
    return e_split, e_loop


def test_stage0_split_vs_fp64(env):
    """Full score table of the split kernel vs fp64, next to the k-ascending fp32 VALU kernel (FLMR_S0_IMPL=valu)."""
    torch, nat, scorer, arrays = env["torch"], env["nat"], env["scorer"], env["arrays"]
    Q = _query_family(torch, env["corpus"], env["synth"])
    C64 = arrays.centroids.astype(np.float64)
    for b in range(Q.size(0)):
        truth = C64 @ Q[b].cpu().numpy().astype(np.float64).T                     # [K, 32]
        tabs = {}
        for impl in ("f16", "valu"):
            with nat.options(FLMR_S0_IMPL=impl):
                scorer.search_batch(Q[b:b + 1], 16, 2, 0.45, 64, 32, full_table=True)
                tabs[impl] = scorer.tap(nat.TAP_CENTROID_SCORES)
        e_split, e_loop = _errs(tabs["f16"], tabs["valu"], truth)
        scale = float(np.abs(truth).max())
        assert e_split.max() <= max(e_loop.max(), 2.0 ** -22 * scale) + FLOOR, (b, e_split.max(), e_loop.max())
        assert e_split.mean() <= e_loop.mean() + FLOOR, (b, e_split.mean(), e_loop.mean())
        assert e_split.max() <= 4e-7 * max(scale, 1.0)                              # fp32-roundoff class in absolute terms


def test_stage2_recompute_vs_fp64(env):
    """Stage-2 document scores of the recomputing MFMA kernel (read back through the phased protocol's stage-2 keys) vs
    fp64: sum_k max_t C[code_t] . q_k over ALL tokens (filter_pids.cpp:27-69 with idx == all ones)."""
    torch, nat, ops, scorer, arrays = env["torch"], env["nat"], env["ops"], env["scorer"], env["arrays"]
    Q = _query_family(torch, env["corpus"], env["synth"])[:2]                      # unit and |q| = 16
    ndocs = 256
    k1 = scorer.phase1(Q, 10, 2, 0.45, ndocs, 32)
    s1 = ops.topn_keys(k1, ndocs, ordered=False)
    k2 = scorer.phase2(s1).cpu().numpy().view(np.uint64)
    C64 = arrays.centroids.astype(np.float64)
    C32 = arrays.centroids
    off = arrays.doc_offsets
    worst = 0.0
    for b in range(Q.size(0)):
        q64 = Q[b].cpu().numpy().astype(np.float64)
        q32 = Q[b].cpu().numpy()
        cs64 = C64 @ q64.T
        cs32 = np.zeros((C32.shape[0], 32), dtype=np.float32)                      # k-ascending fp32 loop
        for d in range(128):
            cs32 += C32[:, d:d + 1] * q32[None, :, d]
        keys = k2[b][k2[b] != 0]
        assert len(keys) >= 64
        pids = (keys & np.uint64(0xFFFFFFFF)).astype(np.int64)
        o = (keys >> np.uint64(32)).astype(np.uint32)
        got = np.where(o & np.uint32(0x80000000), o & np.uint32(0x7FFFFFFF), ~o).astype(np.uint32).view(np.float32)
        e_s, e_l = [], []
        for pid, g in zip(pids, got):
            codes = arrays.codes[off[pid]:off[pid + 1]]
            truth = cs64[codes].max(axis=0).sum()
            loop = np.float32(0)
            for v in cs32[codes].max(axis=0):
                loop = np.float32(loop + v)
            e_s.append(abs(float(g) - truth)), e_l.append(abs(float(loop) - truth))
        scale = max(1.0, float(np.abs(cs64).max()) * 32)
        assert max(e_s) <= max(max(e_l), 2.0 ** -21 * scale) + 32 * FLOOR, (b, max(e_s), max(e_l))
        assert np.mean(e_s) <= np.mean(e_l) * 1.05 + 32 * FLOOR, (b, np.mean(e_s), np.mean(e_l))
        worst = max(worst, max(e_s) / scale)
    assert worst <= 1e-6


def _maxsim_fp64(arrays, pids, q64):
    from ravqa_amd import codec_tables
    rev, lut = codec_tables(arrays.nbits)
    w = arrays.bucket_weights.astype(np.float64)[lut[rev]]                          # [256, 8/nbits]
    out = []
    for pid in pids:
        a, b = arrays.doc_offsets[pid], arrays.doc_offsets[pid + 1]
        D = w[arrays.residuals[a:b]].reshape(b - a, -1) + arrays.centroids[arrays.codes[a:b]].astype(np.float64)
        D /= np.maximum(np.linalg.norm(D, axis=1, keepdims=True), 1e-12)
        sc = D @ q64.T
        out.append(np.maximum(sc.max(axis=0), 0.0).sum() if b > a else 0.0)
    return np.asarray(out)


@pytest.mark.parametrize("nq", [32, 96])
def test_fused_maxsim_split_vs_fp64(env, nq):
    """flmr_score_pids (decompress -> normalise -> split MFMA -> zero-clamped max -> sum) vs the same chain in fp64, next to
    the all-fp32 sequential oracle; Nq = 32 runs the single-tile kernel, Nq = 96 the LDS-chunked long-query kernel."""
    from oracle import oracle as orc
    torch, nat, scorer, arrays = env["torch"], env["nat"], env["scorer"], env["arrays"]
    oi = orc.OracleIndex(arrays.dim, arrays.nbits, arrays.codes, arrays.residuals, arrays.doclens, arrays.ivf,
                         arrays.ivf_lengths, arrays.centroids, arrays.bucket_weights)
    Qs, _ = env["synth"].make_queries(env["corpus"], 3, nq, seed=5)
    Qs = Qs.clone()
    Qs[1] *= 16.0
    Qs[2] *= torch.logspace(-3, 0.9, nq, device="cuda").unsqueeze(1)
    pids = np.arange(100, 400, dtype=np.int32)
    pd = torch.from_numpy(pids).cuda()
    for b in range(3):
        Qd = Qs[b].contiguous()
        out = torch.empty(len(pids), dtype=torch.float32, device="cuda")
        nat.check(scorer._lib.flmr_score_pids(scorer.device_index.handle, C.c_void_p(Qd.data_ptr()), nq, C.c_void_p(pd.data_ptr()),
                                              len(pids), C.c_void_p(out.data_ptr()), nat.stream_ptr()))
        got = out.cpu().numpy()
        qh = Qd.cpu().numpy()
        truth = _maxsim_fp64(arrays, pids, qh.astype(np.float64))
        loop = orc.maxsim_packed(orc.normalize_rows(oi.decompress(pids)), qh, oi.doclens[pids])
        e_s, e_l = _errs(got, loop, truth)
        scale = max(1.0, float(np.abs(truth).max()))
        assert e_s.max() <= max(e_l.max(), 2.0 ** -21 * scale) + nq * FLOOR, (b, e_s.max(), e_l.max())
        # (decompress and the L2 normalisation are fp32 in both chains and dominate the mean; the dot products are what differs)
        assert e_s.mean() <= e_l.mean() * 1.5 + nq * FLOOR, (b, e_s.mean(), e_l.mean())


def test_padded_scorer_split_vs_fp64(env):
    from oracle import oracle as orc
    torch, ops = env["torch"], env["ops"]
    g = torch.Generator().manual_seed(9)
    B, Ld, Nq = 64, 57, 40
    D = torch.nn.functional.normalize(torch.randn(B, Ld, 128, generator=g), dim=-1)
    lens = torch.randint(1, Ld + 1, (B,), generator=g)
    mask = torch.arange(Ld).unsqueeze(0) < lens.unsqueeze(1)
    for scale_q in (1.0, 16.0, 3e-5):
        Q = torch.nn.functional.normalize(torch.randn(1, Nq, 128, generator=g), dim=-1) * scale_q
        got = ops.colbert_score_padded(Q, D, mask).cpu().numpy()
        loop = orc.colbert_score_padded(Q.numpy(), D.numpy(), mask.numpy())
        sc = np.einsum("bld,qd->blq", D.numpy().astype(np.float64), Q[0].numpy().astype(np.float64))
        sc[~mask.numpy()] = -9999.0
        truth = sc.max(axis=1).sum(axis=-1)
        e_s, e_l = _errs(got, loop, truth)
        s = max(float(np.abs(truth).max()), 1e-30)
        assert e_s.max() <= max(e_l.max(), 2.0 ** -21 * s) + Nq * FLOOR, (scale_q, e_s.max(), e_l.max())
        # (the fp32 sum over the Nq column maxima is common to both chains and dominates the mean)
        assert e_s.mean() <= e_l.mean() * 1.5 + Nq * FLOOR, (scale_q, e_s.mean(), e_l.mean())


def test_threshold_within_one_ulp_of_table_maxima(env):
    """thr placed ON a centroid's maximum, one ulp above and one ulp below: the idx bits are exactly `table.max(-1) >= thr`
    on the GPU's own table (the `>=` of index_storage.py:116), the sparse-table path agrees with the full-table path bit for
    bit, and the whole ranking equals the oracle's pruning run on that same table."""
    from oracle import oracle as orc
    torch, nat = env["torch"], env["nat"]
    from ravqa_amd.scorer import IndexScorer
    z = load_golden("idx_nb2")
    import ravqa_amd
    scorer = IndexScorer(arrays=ravqa_amd.IndexArrays.from_golden(z), max_batch=4)
    oi = orc.OracleIndex.from_golden(z)
    Q = torch.from_numpy(z["rank0.Q"]).unsqueeze(0)
    scorer.search_batch(Q, 64, 2, 0.45, 256, 32, full_table=True)
    table = scorer.tap(nat.TAP_CENTROID_SCORES)
    rowmax = table.max(axis=1)
    order = np.argsort(rowmax)
    picks = [rowmax[order[-3]], rowmax[order[-8]], rowmax[order[len(order) // 2]]]   # two survivors' maxima, one mid-table
    for v in picks:
        for thr in (np.nextafter(np.float32(v), np.float32(-np.inf)), np.float32(v), np.nextafter(np.float32(v), np.float32(np.inf))):
            res = {}
            for full in (True, False):
                p, s, c = scorer.search_batch(Q, 64, 2, float(thr), 256, 32, full_table=full)
                bits = scorer.tap(nat.TAP_IDX_BITS)
                idx = np.unpackbits(bits.view(np.uint8), bitorder="little")[: table.shape[0]].astype(bool)
                assert np.array_equal(idx, rowmax >= thr), (float(thr), full)
                res[full] = (p.cpu().numpy().copy(), s.cpu().numpy().copy(), int(c[0]), np.sort(scorer.tap(nat.TAP_STAGE1)),
                             scorer.tap(nat.TAP_STAGE2))
            for a, b in zip(res[True], res[False]):
                assert np.array_equal(a, b), float(thr)
            cand = scorer.tap(nat.TAP_CANDIDATES)
            fin = oi.filter_pids(cand, table, rowmax >= thr, 256)          # the oracle's pruning on the GPU's own table
            assert np.array_equal(res[False][4], fin), float(thr)
