"""`-m gpu` regression tests for launch-path changes that went in without their own coverage (round 3's last two kernel
commits), and for limits the advisor flagged.  Small shapes, seconds in all; they run right after the op-level parity tests.

  * stage 3 reading stage 0's query images (nq <= 32 <= nq_cand): per-query lengths, garbage past a query's length, stale
    images of an earlier, larger batch in the same workspace slots -- against the CPU oracle (TPC/searcher.py:120-126,
    index_storage.py:77: the reference ranks Q[:q_len] and nothing else);
  * s0_prepare_kernel's "hi first" bounds: q_err[k] >= |c . q_lo,k| / 2048 + half an ulp of the score for EVERY centroid, on
    large-magnitude, un-normalised and near-subnormal query rows; q_err_sum >= the sum of the columns' bounds;
  * ndocs = 8192 (the API maximum) through the approximate-then-refine stage 2 (64 KB of dynamic LDS in its plan kernel).
"""
import contextlib

import numpy as np
import pytest

from conftest import tie_aware_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    import ravqa_amd
    from ravqa_amd import _native
    assert torch.cuda.is_available(), "these tests need the MI355X"
    _native.load(require_device=True)
    return dict(torch=torch, pkg=ravqa_amd, native=_native)


def _oracle(corpus):
    from oracle import oracle as orc
    from ravqa_amd import synth
    a = synth.corpus_to_arrays(corpus)
    return orc.OracleIndex(a.dim, a.nbits, a.codes, a.residuals, a.doclens, a.ivf, a.ivf_lengths, a.centroids, a.bucket_weights)


@pytest.mark.parametrize("nq", [32, 20, 7])
def test_stage3_reuses_stage0_images_with_per_query_lengths(hip, nq):
    torch = hip["torch"]
    from ravqa_amd import synth
    from ravqa_amd.scorer import IndexScorer
    from ravqa_amd.searcher import Searcher
    corpus = synth.make_corpus(6000, (3, 70), 2048, 4, seed=31, device="cuda")
    oi = _oracle(corpus)
    scorer = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=32)
    ncells, thr, ndocs = 2, 0.45, 256
    # an earlier, larger batch: 32 long, loud queries leave their images in every workspace slot
    Qa, _ = synth.make_queries(corpus, 32, 32, seed=5)
    scorer.search_batch(Qa * 16.0, ndocs // 4, ncells, thr, ndocs, 32)
    scorer.check()
    # the batch under test: fewer queries, ragged lengths, rows past a query's length hold garbage the reference never sees
    Qb, _ = synth.make_queries(corpus, 13, nq, seed=6)
    lens = [nq, 1, max(1, nq // 2), nq - 1, nq, 3 % nq + 1, nq, 0, nq, max(1, nq - 5), 2 % nq + 1, nq, nq // 3 + 1]
    g = torch.Generator(device="cuda").manual_seed(9)
    Qdirty = Qb.clone()
    for i, n in enumerate(lens):
        Qdirty[i, n:] = torch.randn((nq - n, 128), generator=g, device="cuda") * 3.0
    q_lens = torch.tensor(lens, dtype=torch.int32)
    p, s, c = scorer.search_batch(Qdirty, ndocs // 4, ncells, thr, ndocs, 32, q_lens=q_lens)
    scorer.check()
    p, s, c = p.cpu().numpy(), s.cpu().numpy(), c.cpu().numpy()
    Qh = Qb.cpu().numpy()
    checked = 0
    for i, n in enumerate(lens):
        if n == 0:      # an empty query: defined here (every candidate scores 0), the reference never ranks one
            assert np.all(s[i, : int(c[i])] == 0.0)
            continue
        rp, rs, ncand = oi.rank(Qh[i, :n], ncells, thr, ndocs, 32)
        if ncand < ndocs:
            continue
        m = int(c[i])
        assert m == len(rp), (i, n, m, len(rp))
        tie_aware_equal(rp, rs, p[i, :m], s[i, :m])
        checked += 1
    assert checked >= 8, checked
    # remove_zero_tensors (searcher.py:120-126): zero rows in the MIDDLE of a query, compacted on the host, then the same path
    Qz = Qb.clone()
    Qz[:, 1::3] = 0.0
    Qc, zl = Searcher._compact_nonzero_rows(Qz)
    p2, s2, c2 = scorer.search_batch(Qc, ndocs // 4, ncells, thr, ndocs, 32, q_lens=zl)
    scorer.check()
    for i in (0, 4, 8):
        keep = Qz[i].abs().sum(-1) != 0
        rp, rs, ncand = oi.rank(Qz[i][keep].cpu().numpy(), ncells, thr, ndocs, 32)
        if ncand >= ndocs:
            m = int(c2[i])
            tie_aware_equal(rp, rs, p2[i, :m].cpu().numpy(), s2[i, :m].cpu().numpy())
    scorer.close_searcher()


def test_hi_first_bounds_cover_the_lo_product_on_adversarial_queries(hip):
    """q_err[k] must bound |s - a_hi| = |fl(a_hi + a_lo / 2048) - a_hi| for every centroid c: |a_lo| / 2048 + ulp/2(s), with
    a_lo = c . q_lo (the fp16 image of 2048 (q - q_hi)) evaluated here in fp64 on the same images."""
    torch, nat = hip["torch"], hip["native"]
    from ravqa_amd import synth
    from ravqa_amd.scorer import IndexScorer
    K = 4096
    corpus = synth.make_corpus(3000, (4, 40), K, 2, seed=41, device="cuda")
    # a table with LARGE rows too (the bound scales with the largest centroid norm of the index)
    cen = corpus.centroids.clone()
    cen[100:200] *= 7.5
    cen[300] = 0.0
    corpus.centroids = cen.half().float().contiguous()
    scorer = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=16)
    g = torch.Generator(device="cuda").manual_seed(3)
    Q, _ = synth.make_queries(corpus, 8, 32, seed=8)
    Q[1] *= 16.0                                                           # FLMR's un-normalised visual tokens
    Q[2] *= 1000.0                                                         # far outside anything trained
    Q[3] *= torch.logspace(-6, 1.5, 32, device="cuda").unsqueeze(1)        # rows whose hi image is subnormal in fp16
    Q[4] = torch.randn((32, 128), generator=g, device="cuda") * 1e-7      # every lo image underflows
    Q[5] = (torch.rand((32, 128), generator=g, device="cuda") < 0.5).float() * (1.0 + 2.0 ** -12)   # lo = the largest it can be
    Q[6, :, ::2] = 0.0
    Q[7] = Q[0] * 65000.0 / Q[0].abs().max()                               # hi at the top of fp16's range
    scorer.search_batch(Q, 16, 2, 0.45, 64, 32)
    scorer.check()
    C = corpus.centroids.double().cpu().numpy()
    cmax = float(np.sqrt((C ** 2).sum(1)).max())
    for q in range(Q.size(0)):
        err = scorer.tap(nat.TAP_Q_ERR, q)
        esum = scorer.tap(nat.TAP_Q_ERR_SUM, q)
        assert err.shape == (32,) and esum.shape == (1,), "the default path of this shape is hi first"
        v = Q[q].cpu().numpy().astype(np.float32)
        with np.errstate(over="ignore"):
            hi = v.astype(np.float16)
            lo = ((v - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
        assert np.all(np.isfinite(hi.astype(np.float32))), q
        a_lo = C @ lo.astype(np.float64).T                                 # [K, 32]
        s_full = C @ (hi.astype(np.float64) + lo.astype(np.float64) / 2048.0).T
        need = np.abs(a_lo) / 2048.0 + np.abs(s_full) * 2.0 ** -24
        worst = need.max(0)
        assert np.all(err.astype(np.float64) >= worst), (q, float((worst - err).max()))
        # not vacuous either: within 64x of the worst centroid for rows that have a lo image at all
        live = worst > 1e-30
        assert np.all(err[live] <= 64.0 * np.maximum(worst[live], cmax * np.abs(v).max() * 1e-7)), q
        assert float(esum[0]) >= float(err.astype(np.float64).sum()), q
    scorer.close_searcher()


def test_ndocs_8192_through_approximate_then_refine(hip):
    """The API maximum ndocs (FLMR_MAX_NDOCS): the refine plan sorts 8192 keys per query in 64 KB of dynamic LDS.  Default path
    vs the full-score sorted form (FLMR_S2_IMPL=xcd) and vs the gather form: identical final output."""
    torch, nat = hip["torch"], hip["native"]
    from ravqa_amd import synth
    from ravqa_amd.scorer import IndexScorer
    corpus = synth.make_corpus(40_000, (8, 40), 8192, 2, seed=51, device="cuda")
    scorer = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=8)
    Q, _ = synth.make_queries(corpus, 6, 32, seed=2)
    res = {}
    for impl in ("xcda", "xcd", "lds"):
        with nat.options(FLMR_S2_IMPL=impl):
            p, s, c = scorer.search_batch(Q, 2048, 8, 0.3, 8192, 32)
            scorer.check()
            res[impl] = (p.cpu().numpy(), s.cpu().numpy().view(np.uint32), c.cpu().numpy())
    assert int(res["lds"][2].min()) >= 1024, res["lds"][2]
    for impl in ("xcda", "xcd"):
        for x, y in zip(res[impl], res["lds"]):
            assert np.array_equal(x, y), impl
    scorer.close_searcher()


def test_many_surviving_centroids_code_scan_and_row_capacity(hip):
    """Thousands of centroids above the threshold per query (what a real corpus gives at the policy thresholds; the synthetic
    corpora have ~50): beyond the scatter stage 1's 1024 lists the code-scanning stage 1 runs, on the compact score rows
    (row = rank of the centroid among the survivors) -- ranked lists against the CPU oracle.  A searcher created with fewer
    score rows than a query needs gives the same results (the library recomputes that query's stage 1)."""
    torch, nat = hip["torch"], hip["native"]
    from ravqa_amd import synth
    from ravqa_amd.scorer import IndexScorer
    corpus = synth.make_corpus(30_000, (20, 60), 8192, 2, seed=61, device="cuda")
    oi = _oracle(corpus)
    scorer = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=16)
    Q, _ = synth.make_queries(corpus, 12, 32, seed=3)
    q_lens = torch.tensor([32, 32, 17, 32, 32, 5, 32, 32, 32, 32, 32, 32], dtype=torch.int32)
    ncells, thr, ndocs = 2, 0.2, 256
    p, s, c = scorer.search_batch(Q, ndocs // 4, ncells, thr, ndocs, 32, q_lens=q_lens)
    scorer.check()
    nsurv = [int(np.unpackbits(scorer.tap(nat.TAP_IDX_BITS, q).view(np.uint8)).sum()) for q in range(Q.size(0))]
    assert max(nsurv) > 1024, nsurv        # the shape under test: past the scatter kernel's list ids
    p, s, c = p.cpu().numpy(), s.cpu().numpy(), c.cpu().numpy()
    Qh = Q.cpu().numpy()
    checked = 0
    for i in range(Q.size(0)):
        rp, rs, ncand = oi.rank(Qh[i, : int(q_lens[i])], ncells, thr, ndocs, 32)
        if ncand < ndocs:
            continue
        m = int(c[i])
        assert m == len(rp), (i, m, len(rp))
        tie_aware_equal(rp, rs, p[i, :m], s[i, :m])
        checked += 1
    assert checked >= 8, checked
    # A searcher that keeps FEWER score rows than a query has surviving centroids: the reference has no such limit
    # (index_storage.py:116), so the library recomputes that query's stage 1 from the centroids inside the same batch
    # (filter_stage1_recompute_kernel) -- no error, no warning, bit for bit the same result, on every path above the C ABI.
    import warnings
    with nat.options(FLMR_ROW_CAP="256"), warnings.catch_warnings():
        warnings.simplefilter("error")
        pf, sf, cf = scorer.search_batch(Q, ndocs // 4, ncells, thr, ndocs, 32, q_lens=q_lens)
        scorer.check()
        forms = [int(scorer.tap(nat.TAP_STAGE1_FORM, q)[0]) for q in range(Q.size(0)) if scorer.tap(nat.TAP_STAGE1_FORM, q).size]
        assert forms.count(7) >= 8, forms       # the recompute form took the queries over the capacity
        assert np.array_equal(pf.cpu().numpy(), p) and np.array_equal(cf.cpu().numpy(), c)
        assert np.array_equal(sf.cpu().numpy().view(np.uint32), s.view(np.uint32))
        pend = scorer.search_batch_pending(Q, ndocs // 4, ncells, thr, ndocs, 32, q_lens=q_lens)
        pend.wait_all()
        assert np.array_equal(pend.pids, p) and np.array_equal(pend.counts, c) and np.array_equal(pend.scores.view(np.uint32), s.view(np.uint32))
        from ravqa_amd.distributed import ShardedSearcher
        ss = ShardedSearcher(scorer=scorer, k_policy=lambda k_: (ncells, thr, ndocs))
        pz, sz, cz = ss.search_batch(Q, ndocs // 4, q_lens=q_lens)
        assert np.array_equal(pz.cpu().numpy(), p) and np.array_equal(sz.cpu().numpy().view(np.uint32), s.view(np.uint32))
        pe, se, ce = ss.search_batch_exact(Q, ndocs // 4, q_lens=q_lens, gather=lambda t: t.unsqueeze(0), reduce_sum=lambda t: t)
        assert np.array_equal(pe.cpu().numpy(), p) and np.array_equal(ce.cpu().numpy(), c)
        assert np.array_equal(se.cpu().numpy().view(np.uint32), s.view(np.uint32))
        p2, s2, c2 = scorer.search_batch(Q, ndocs // 4, ncells, 0.6, ndocs, 32, q_lens=q_lens)   # few survivors: the usual forms
        scorer.check()
        p3, s3, c3 = IndexScorer(device_index=scorer.device_index, max_batch=16).search_batch(Q, ndocs // 4, ncells, 0.6, ndocs, 32, q_lens=q_lens)
        assert torch.equal(p2, p3) and torch.equal(s2, s3) and torch.equal(c2, c3)
    scorer.close_searcher()


def test_build_ivf_matches_the_reference_written_ivf(hip):
    """flmr_build_ivf (stable device radix sort of (code, pid) by code, run flags, compaction) against the IVF the REFERENCE's
    optimize_ivf wrote for the golden indexes (TPC/indexing/utils.py:8-53), bit for bit, and against the torch.unique
    restatement on a corpus with empty passages, a code that never occurs and 2.6 M tokens."""
    torch = hip["torch"]
    from conftest import INDEX_FIXTURES, load_golden
    from ravqa_amd import ops, synth
    for name in INDEX_FIXTURES:
        z = load_golden(name)
        a = hip["pkg"].IndexArrays.from_golden(z)
        ivf, lens = ops.build_ivf(torch.from_numpy(a.codes), torch.from_numpy(a.doclens), a.num_centroids)
        assert np.array_equal(ivf.cpu().numpy(), a.ivf) and np.array_equal(lens.cpu().numpy(), a.ivf_lengths), name
    g = torch.Generator(device="cuda").manual_seed(5)
    P, K = 40_000, 3000
    doclens = torch.randint(0, 130, (P,), generator=g, device="cuda")
    codes = torch.randint(0, K - 1, (int(doclens.sum()),), generator=g, device="cuda", dtype=torch.int32)   # code K - 1 never occurs
    ivf, lens = ops.build_ivf(codes, doclens, K)
    rivf, rlens = synth.build_ivf(codes, doclens, K)
    assert torch.equal(ivf, rivf) and torch.equal(lens, rlens) and int(lens[K - 1]) == 0
    e_ivf, e_lens = ops.build_ivf(codes[:0], doclens[:0], 8)
    assert e_ivf.numel() == 0 and int(e_lens.sum()) == 0


def test_flipr_interaction_through_the_scoring_dispatch(hip):
    """colbert_score with config.interaction == 'flipr' (TPC/modeling/colbert.py:246-261: the 32 largest column maxima of the
    first 64 query tokens + the 8 largest of the rest) reaches the HIP scorer too: flmr_colbert_colmax_padded returns the
    column maxima, the top-k sums are taken on the device -- against the reference's expression restated in torch fp64."""
    torch = hip["torch"]
    from types import SimpleNamespace
    from ravqa_amd import ops
    from ravqa_amd.dropin import make_colbert_score_dispatch
    g = torch.Generator().manual_seed(4)
    B, Ld, Nq = 37, 45, 80
    Q = torch.nn.functional.normalize(torch.randn(1, Nq, 128, generator=g), dim=-1)
    D = torch.nn.functional.normalize(torch.randn(B, Ld, 128, generator=g), dim=-1)
    mask = torch.rand(B, Ld, generator=g) < 0.8
    mask[:, 0] = True

    def reference(Q, D_padded, D_mask, config=None, use_gpu=False):   # colbert.py:235-286 in fp64
        sc = D_padded.double() @ Q.double().permute(0, 2, 1)
        sc[~D_mask.view(sc.size(0), sc.size(1)).bool()] = -9999
        cm = sc.max(1).values
        if config is not None and config.interaction == "flipr":
            out = cm[:, :config.query_maxlen].topk(config.query_maxlen // 2, dim=-1).values.sum(-1)
            if 8 <= cm.size(1) - config.query_maxlen:
                out = out + cm[:, config.query_maxlen:].topk(8, dim=-1).values.sum(1)
            return out
        return cm.sum(-1)

    calls = []
    fn = make_colbert_score_dispatch(lambda *a, **k: (calls.append(1), reference(*a, **k))[1])
    for nq in (Nq, 64, 70):     # with, without and with too few extra tokens for the second term
        for cfg in (SimpleNamespace(interaction="flipr", query_maxlen=64), SimpleNamespace(interaction="colbert", query_maxlen=64)):
            with torch.no_grad():
                got = fn(Q[:, :nq], D, mask.unsqueeze(-1), config=cfg)
            want = reference(Q[:, :nq], D, mask, config=cfg)
            assert got.shape == (B,) and got.dtype == D.dtype and float((got.double() - want).abs().max()) <= 1e-4, (nq, cfg.interaction)
    assert not calls      # every call above ran on the HIP scorer
    cm = ops.colbert_colmax_padded(Q, D, mask).cpu()
    sc = D.double() @ Q.double().permute(0, 2, 1)
    sc[~mask] = -9999
    assert float((cm.double() - sc.max(1).values).abs().max()) <= 1e-5


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_dense_survivor_sets_vs_oracle(hip, seed):
    """Seeded shapes at LOW thresholds -- hundreds to thousands of centroids above centroid_score_threshold per query, most
    stage-0 tiles flagged, the idx words decided inline from the hi products, compact score rows by rank, the code-scanning
    stage 1 for the queries past the scatter form's limits and the scatter form for the others IN THE SAME BATCH -- against
    the oracle.  Every nbits, ragged passages with empties, per-query lengths, K any multiple of 128, passages longer than one
    128-token chunk in some cases."""
    torch, nat = hip["torch"], hip["native"]
    from ravqa_amd import synth
    from ravqa_amd.scorer import IndexScorer
    rng = np.random.default_rng(9000 + seed)
    K = 128 * int(rng.integers(8, 96))
    nbits = int(rng.choice([1, 2, 4, 8]))
    npass = int(rng.choice([4000, 12000, 30000]))
    lo = int(rng.integers(0, 10))
    doclen = (lo, lo + int(rng.choice([20, 60, 150, 300])))
    nqueries = int(rng.integers(9, 40))
    nq = int(rng.choice([32, 32, 20]))
    ncells = int(rng.integers(1, 4))
    thr = float(rng.choice([0.08, 0.12, 0.18, 0.25]))
    ndocs = int(rng.choice([64, 256]))
    corpus = synth.make_corpus(npass, doclen, K, nbits, seed=700 + seed, device="cuda")
    oi = _oracle(corpus)
    Q, _ = synth.make_queries(corpus, nqueries, nq, seed=800 + seed)
    Q[1::4] *= float(rng.choice([0.5, 2.0, 16.0]))        # some queries with far fewer / far more survivors than the others
    q_lens = torch.from_numpy(rng.integers(1, nq + 1, size=nqueries).astype(np.int32))
    q_lens[::3] = nq
    scorer = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=64)
    p, s, c = scorer.search_batch(Q, ndocs // 4, ncells, thr, ndocs, 32, q_lens=q_lens)
    scorer.check()
    p, s, c = p.cpu().numpy(), s.cpu().numpy(), c.cpu().numpy()
    Qh = Q.cpu().numpy()
    nsurv, checked = [], 0
    for i in list(range(0, nqueries, max(1, nqueries // 8)))[:8]:
        nsurv.append(int(np.unpackbits(scorer.tap(nat.TAP_IDX_BITS, i).view(np.uint8)).sum()))
        ql = int(q_lens[i])
        rp, rs, ncand = oi.rank(Qh[i, :ql], ncells, thr, ndocs, 32)
        n = int(c[i])
        if ncand < ndocs:
            assert n == min(ncand, ndocs // 4), (seed, i, n, ncand)
            continue
        mag = max(1.0, float(np.max(np.abs(rs))))
        tie_aware_equal(rp, rs, p[i, :n], s[i, :n], gap=max(1e-5, 4 * 1.2e-7 * mag), tol=max(1e-4, 16 * 1.2e-7 * mag))
        checked += 1
    assert max(nsurv) >= 64, (seed, nsurv)
    scorer.close_searcher()


def test_long_query_kernel_scales_per_query(hip):
    """The query-stationary long-query S3 kernel (Nq >= 288) scales its fp16 operands per query: queries whose rows are far from
    unit norm (FLMR's un-normalised visual tokens; here x 700 and x 1e-3) must score like the chunked kernel (FLMR_S3_IMPL=regs,
    whose split has no such scale) -- relative to the scores' magnitude."""
    import ctypes as C
    nat, torch = hip["native"], hip["torch"]
    from ravqa_amd import synth
    from ravqa_amd.scorer import IndexScorer
    corpus = synth.make_corpus(4000, (5, 150), 512, 2, seed=5, device="cuda")
    scorer = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=4)
    Q, _ = synth.make_queries(corpus, 1, 320, seed=4)
    pids = torch.arange(0, 4000, 7, dtype=torch.int32, device="cuda")
    for scale in (1.0, 700.0, 1e-3):
        Qd = (Q[0] * scale).contiguous()
        outs = {}
        for impl in ("regs", "qs"):
            with nat.options(FLMR_S3_IMPL=impl):
                out = torch.empty(pids.numel(), dtype=torch.float32, device="cuda")
                nat.check(scorer._lib.flmr_score_pids(scorer.device_index.handle, C.c_void_p(Qd.data_ptr()), 320, C.c_void_p(pids.data_ptr()),
                                                      pids.numel(), C.c_void_p(out.data_ptr()), nat.stream_ptr()))
                outs[impl] = out.cpu().numpy().astype(np.float64)
        assert np.all(np.isfinite(outs["qs"])), scale
        mag = float(np.max(np.abs(outs["regs"]))) + 1e-30
        assert np.max(np.abs(outs["qs"] - outs["regs"])) <= 2e-6 * mag, (scale, np.max(np.abs(outs["qs"] - outs["regs"])), mag)
    scorer.close_searcher()


@pytest.mark.parametrize("policy,numerics", [((2, 0.45, 1024), "cpu"), ((4, 0.4, 4096), "cpu"), ((8, 0.38, 1024), "cpu"), ((8, 0.35, 1024), "cpu"),
                                             ((8, 0.33, 1024), "cpu"), ((2, 0.45, 1024), "gpu-fp16"), ((8, 0.35, 1024), "gpu-fp16")])
def test_stage1_queue_form_equals_slot_form_and_code_scan(hip, policy, numerics):
    """The list-scatter stage 1 has two forms: the queue form (cand_fast_kernel: one barrier per chunk, single-centroid
    passages scored by their list's constant, the pairs of the others queued) runs first and hands the queries it cannot
    finish to the slot form (cand_mark_score_kernel).  On a corpus of the bench's shape (hundreds of thousands of passages,
    K in the tens of thousands: ~30 entries per list and 32768-passage chunk) nearly every query must stay in the queue form, and
    its survivors, their order after the selection, and the final ranking must be IDENTICAL to the slot form's and to the
    code-scanning stage 1 (filter_pids.cpp:27-69).  Ragged lengths and an empty query included.  The third policy (ncells = 8,
    threshold 0.38) gives a wave more lists than it keeps in registers across the barrier (16 probed + 8 surviving): the reloading
    loops of the mark and pair passes run."""
    torch, nat = hip["torch"], hip["native"]
    from ravqa_amd import synth
    from ravqa_amd.scorer import IndexScorer
    ncells, thr, ndocs = policy
    corpus = synth.make_corpus(300_000, (16, 112), 65536, 2, seed=21, device="cuda")
    nqueries = 40
    Q, _ = synth.make_queries(corpus, nqueries, 32, seed=22)
    q_lens = torch.full((nqueries,), 32, dtype=torch.int32)
    q_lens[3], q_lens[7], q_lens[11] = 9, 0, 31
    scorer = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=64, numerics=numerics)   # (gpu-fp16: the kernels' F16 instantiations)
    assert scorer.numerics == numerics
    outs, forms = {}, None
    for tag, env in (("queue", {}), ("slots", {"FLMR_S1_IMPL": "slots"}), ("scan", {"FLMR_S1_IMPL": "scan"})):
        with nat.options(**env):
            Qw, _ = synth.make_queries(corpus, 8, 32, seed=23)
            scorer.search_batch(Qw, ndocs // 4, ncells, thr, ndocs, 32)   # (no form may live off the previous one's workspace)
            p, s, c = scorer.search_batch(Q, ndocs // 4, ncells, thr, ndocs, 32, q_lens=q_lens)
            scorer.check()
            if tag == "queue":
                forms = [int(scorer.tap(nat.TAP_STAGE1_FORM, i)[0]) for i in range(nqueries)]
            else:
                assert scorer.tap(nat.TAP_STAGE1_FORM, 0).size == 0   # (the queue form did not run)
            outs[tag] = ([np.sort(scorer.tap(nat.TAP_STAGE1, i)) for i in range(nqueries)],
                         [scorer.tap(nat.TAP_STAGE2, i) for i in range(nqueries)], p.cpu().numpy(), s.cpu().numpy(), c.cpu().numpy())
    nsurv = [int(np.unpackbits(scorer.tap(nat.TAP_IDX_BITS, i).view(np.uint8)).sum()) for i in range(nqueries)]
    ncell = [int(scorer.tap(nat.TAP_CELLS, i).size) for i in range(nqueries)]
    print("\nSUMMARY policy", policy, "forms", [forms.count(v) for v in (0, 1, 2)], "surviving centroids median", int(np.median(nsurv)), "max", max(nsurv),
          "cells max", max(ncell))
    if ncells == 8:   # (a low threshold: some queries are past the queue form's limits; enough must stay that run the reloading loops)
        print("reloading loops:", sum(1 for f, n, c in zip(forms, nsurv, ncell) if f == 0 and n > 64 and c > 128), "of", nqueries)
    else:
        assert sum(1 for f in forms if f == 0) >= nqueries - 4, forms
    for tag in ("slots", "scan"):
        for i in range(nqueries):
            assert np.array_equal(outs["queue"][0][i], outs[tag][0][i]), (tag, i)
            assert np.array_equal(outs["queue"][1][i], outs[tag][1][i]), (tag, i)
        for a, b in zip(outs["queue"][2:], outs[tag][2:]):
            assert np.array_equal(a, b), tag
    scorer.close_searcher()


def test_stage1_forms_are_planned_per_query_on_a_built_index(hip):
    """An index BUILT from overlapping clusters: every hit passage of a full query holds several surviving centroids -- the queue
    form of the list scatter would overflow at once -- so cand_plan_kernel, which MEASURES hit candidates and queued pairs on a sample
    of the query's chunks, plans the small-dense form (3) for them from the FIRST batch on (no searcher history: the plan is a
    function of the query and the index), while the short queries of the same batch (few surviving lists, few pairs) go to the
    queue form (0) or, where their surviving lists are far longer than their probed cells', to the dense image form (5).  One batch,
    at least three forms, and survivors / finalists / results IDENTICAL to the slot form's and the code scan's."""
    torch, nat = hip["torch"], hip["native"]
    from ravqa_amd import indexing, synth
    from ravqa_amd.scorer import IndexScorer
    embs, doclens, planted = synth.make_overlapping_embeddings(300_000, 64, 2048, seed=5, device="cuda", sub_directions=8192)
    arrays = indexing.build_index(embs, doclens, nbits=2, kmeans_niters=4)
    ncells, thr, ndocs = 2, 0.45, 1024
    Q = planted(64)[0]
    Q2 = planted(64)[0]
    for j in range(48, 64):      # queries whose tokens come from 8 DIFFERENT planted queries: more topics, many more hit candidates per chunk
        for t in range(32):
            Q[j, t] = Q2[(j + 5 * (t // 4)) % 64, t]
    q_lens = torch.full((64,), 32, dtype=torch.int32)
    q_lens[32:48] = 1
    scorer = IndexScorer(arrays=arrays, max_batch=64)
    outs, forms = {}, None
    for tag, env in (("planned", {}), ("slots", {"FLMR_S1_IMPL": "slots"}), ("scan", {"FLMR_S1_IMPL": "scan"})):
        with nat.options(**env):
            p, s, c = scorer.search_batch(Q, ndocs // 4, ncells, thr, ndocs, 32, q_lens=q_lens)   # (the first batch of a fresh searcher)
            scorer.check()
            if tag == "planned":
                forms = [int(scorer.tap(nat.TAP_STAGE1_FORM, i)[0]) for i in range(64)]
            outs[tag] = ([np.sort(scorer.tap(nat.TAP_STAGE1, i)) for i in range(64)], [scorer.tap(nat.TAP_STAGE2, i) for i in range(64)],
                         p.cpu().numpy(), s.cpu().numpy(), c.cpu().numpy())
    nsurv = [int(np.unpackbits(scorer.tap(nat.TAP_IDX_BITS, i).view(np.uint8)).sum()) for i in range(0, 64, 8)]
    print("\nFORMS full", {v: forms[:32].count(v) for v in sorted(set(forms[:32]))}, "one token", {v: forms[32:48].count(v) for v in sorted(set(forms[32:48]))},
          "8 topics", {v: forms[48:].count(v) for v in sorted(set(forms[48:]))}, "surviving centroids", nsurv)
    assert forms[:32].count(3) >= 24, forms          # planned for the small-dense form, and finished by it
    assert len(set(forms)) >= 3, forms               # one batch, three forms
    for tag in ("slots", "scan"):
        for i in range(64):
            assert np.array_equal(outs["planned"][0][i], outs[tag][0][i]) and np.array_equal(outs["planned"][1][i], outs[tag][1][i]), (tag, i, forms[i])
        for x, y in zip(outs["planned"][2:], outs[tag][2:]):
            assert np.array_equal(x, y), tag
    scorer.close_searcher()


def _dense_vs_scan(hip, corpus, policy, nqueries=24):
    """one batch through the default path and through FLMR_S1_IMPL=scan: identical survivors, finalists, results; returns the forms"""
    torch, nat = hip["torch"], hip["native"]
    from ravqa_amd import synth
    from ravqa_amd.scorer import IndexScorer
    Q, _ = synth.make_queries(corpus, nqueries, 32, seed=78)
    q_lens = torch.full((nqueries,), 32, dtype=torch.int32)
    q_lens[2], q_lens[5], q_lens[9] = 11, 0, 1
    scorer = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=32)
    ncells, thr, ndocs = policy
    outs, forms = {}, None
    for tag, env in (("dense", {}), ("scan", {"FLMR_S1_IMPL": "scan"})):
        with nat.options(**env):
            p, s, c = scorer.search_batch(Q, ndocs // 4, ncells, thr, ndocs, 32, q_lens=q_lens)
            scorer.check()
            if tag == "dense":
                forms = [int(scorer.tap(nat.TAP_STAGE1_FORM, i)[0]) if scorer.tap(nat.TAP_STAGE1_FORM, i).size else -1 for i in range(nqueries)]
            outs[tag] = ([np.sort(scorer.tap(nat.TAP_STAGE1, i)) for i in range(nqueries)],
                         [scorer.tap(nat.TAP_STAGE2, i) for i in range(nqueries)], p.cpu().numpy(), s.cpu().numpy(), c.cpu().numpy())
    nsurv = [int(np.unpackbits(scorer.tap(nat.TAP_IDX_BITS, i).view(np.uint8)).sum()) for i in range(nqueries)]
    print("\nPOLICY", policy, "FORMS", {v: forms.count(v) for v in sorted(set(forms))}, "surviving centroids min / median / max", min(nsurv), int(np.median(nsurv)), max(nsurv))
    for i in range(nqueries):
        assert np.array_equal(outs["dense"][0][i], outs["scan"][0][i]), ("stage-1 survivors", policy, i, forms[i], nsurv[i])
        assert np.array_equal(outs["dense"][1][i], outs["scan"][1][i]), ("stage-2 finalists", policy, i, forms[i], nsurv[i])
    assert np.array_equal(outs["dense"][4], outs["scan"][4]) and np.array_equal(outs["dense"][2], outs["scan"][2])
    assert np.array_equal(outs["dense"][3].view(np.uint32), outs["scan"][3].view(np.uint32))
    scorer.close_searcher()
    return forms


def test_stage1_dense_forms_equal_code_scan(hip):
    """Queries with more surviving centroids than the list-scatter forms take run the dense forms of flmr_stage1_dense.hip: an
    approximate pass on fp16 IMAGES of the score rows (rounded up, kept in LDS), the band of candidates within the images' error of
    the cut, and an exact pass over the band -- or the exact pass over every candidate when the rows do not fit.  The survivors,
    the stage-2 finalists and the final result must be bit for bit those of the code-scanning kernel (FLMR_S1_IMPL=scan:
    filter_pids.cpp:27-124 on every candidate), ragged and empty queries included.  The threshold is lowered step by step so that
    the batch passes from the list-scatter forms through the image form (5) to the exact form (6): both must have been seen."""
    from ravqa_amd import synth
    corpus = synth.make_corpus(30_000, (8, 100), 4096, 2, seed=77, device="cuda")
    seen = set()
    for thr in (0.30, 0.25, 0.20, 0.15, 0.10, 0.05, -0.05):
        seen |= set(_dense_vs_scan(hip, corpus, (2, thr, 1024)))
    assert 5 in seen and 6 in seen, seen


@pytest.mark.parametrize("K,npass,doclen,policy", [
    (8192, 60_000, 64, (2, 0.15, 4096)),             # more survivors wanted than a band would hold back
    (2048, 6_000, (1, 300), (2, 0.15, 64)),          # passages longer than one 64 / 128-code chunk, tiny ndocs
    (2048, 6_000, (1, 300), (4, -1.0, 8192)),        # every centroid survives, every candidate is selected
])
def test_stage1_dense_forms_shapes(hip, K, npass, doclen, policy):
    from ravqa_amd import synth
    corpus = synth.make_corpus(npass, doclen, K, 2, seed=79, device="cuda")
    forms = _dense_vs_scan(hip, corpus, policy)
    assert any(f in (5, 6) for f in forms), forms


def test_scoring_surface_flipr_l2_and_the_cross_form(hip):
    """The scoring head's host surface: `scoring.colbert_score` / `colbert_score_reduce` serve the 'flipr' interaction as the patched
    dispatch does (TPC/modeling/colbert.py:246-261), `FLMRModelForRetrieval.score` with similarity == 'l2' is the reference's
    torch expression (colbert.py:220-222), and the cross form of the padded kernel (every query against every document in one
    launch: the exhaustive search's rate matrix, FLMR_executor.py:799-847) equals one broadcast call per query, bit for bit, on
    both kernels (dim 128: MFMA; dim 64: plain FMA)."""
    torch = hip["torch"]
    from types import SimpleNamespace
    from ravqa_amd import ops, scoring
    from ravqa_amd.flmr import FLMRModelForRetrieval
    g = torch.Generator().manual_seed(9)
    B, Ld, Nq = 29, 37, 80
    Q = torch.nn.functional.normalize(torch.randn(1, Nq, 128, generator=g), dim=-1)
    D = torch.nn.functional.normalize(torch.randn(B, Ld, 128, generator=g), dim=-1)
    mask = torch.rand(B, Ld, generator=g) < 0.8
    mask[:, 0] = True
    cfg = SimpleNamespace(interaction="flipr", query_maxlen=64)
    sc = D.double() @ Q.double().permute(0, 2, 1)
    sc[~mask] = -9999
    cm = sc.max(1).values
    want = cm[:, :64].topk(32, dim=-1).values.sum(-1) + cm[:, 64:].topk(8, dim=-1).values.sum(1)
    with torch.no_grad():
        got = scoring.colbert_score(Q, D, mask, config=cfg).cpu().double()
        got2 = scoring.colbert_score_reduce((D @ Q.permute(0, 2, 1)).cuda(), mask.cuda(), cfg).cpu().double()
        plain = scoring.colbert_score(Q, D, mask, config=SimpleNamespace(interaction="colbert")).cpu().double()
    assert float((got - want).abs().max()) <= 1e-4 and float((got2 - want).abs().max()) <= 1e-4
    assert float((plain - cm.sum(-1)).abs().max()) <= 1e-4
    # similarity 'l2' (no mask upstream either)
    model = FLMRModelForRetrieval(text_encoder=None, colbert_config=SimpleNamespace(similarity="l2", interaction="colbert"))
    Qb = Q.expand(B, -1, -1)
    l2 = model.score(Qb, D, mask)
    ref = (-1.0 * ((Qb.unsqueeze(2) - D.unsqueeze(1)) ** 2).sum(-1)).max(-1).values.sum(-1)
    assert torch.equal(l2, ref)
    # cross form == one broadcast call per query
    for dim in (128, 64):
        Qs = torch.nn.functional.normalize(torch.randn(7, 40, dim, generator=g), dim=-1)
        Ds = torch.nn.functional.normalize(torch.randn(B, Ld, dim, generator=g), dim=-1)
        cross = ops.colbert_score_cross(Qs, Ds, mask).cpu()
        assert cross.shape == (7, B)
        for q in range(7):
            one = ops.colbert_score_padded(Qs[q:q + 1], Ds, mask).cpu()
            assert torch.equal(cross[q].view(torch.int32), one.view(torch.int32)), (dim, q)
