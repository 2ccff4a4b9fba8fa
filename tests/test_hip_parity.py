"""Parity of the HIP path (through the C ABI) against the reference's golden vectors and the CPU oracle.

Bars (north_star): integer / index work bit-exact; fp32 scores within 1e-4; ranked ids identical except inside
runs of reference scores closer than 1e-5 (a different valid fp32 summation order may swap those).
"""
import contextlib
import os

import numpy as np
import pytest

from conftest import INDEX_FIXTURES, load_golden, rank_records, tie_aware_equal

pytestmark = pytest.mark.gpu

GEMM_TOL = 2e-6
SCORE_TOL = 1e-4


@pytest.fixture(scope="module")
def hip():
    import torch
    import ravqa_amd
    from ravqa_amd import _native, ops
    from ravqa_amd.scorer import IndexScorer
    assert torch.cuda.is_available(), "these tests need the MI355X"
    _native.load(require_device=True)  # fails loudly if libflmr_hip.so is missing: no fallback
    return dict(torch=torch, pkg=ravqa_amd, native=_native, ops=ops, IndexScorer=IndexScorer)


@pytest.fixture(scope="module")
def scorers(hip):
    out = {}
    for name in INDEX_FIXTURES:
        z = load_golden(name)
        out[name] = (z, hip["IndexScorer"](arrays=hip["pkg"].IndexArrays.from_golden(z)))
    return out


def _search_one(hip, scorer, z, r, full_table=False):
    torch = hip["torch"]
    Q = torch.from_numpy(z[f"{r}.Q"]).unsqueeze(0)
    ncells, thr, ndocs = int(z[f"{r}.ncells"]), float(z[f"{r}.thr"]), int(z[f"{r}.ndocs"])
    p, s, c = scorer.search_batch(Q, max(ndocs // 4, 1), ncells, thr, ndocs, int(z[f"{r}.nq_cand"]), full_table=full_table)
    n = int(c[0])
    return p[0, :n].cpu().numpy(), s[0, :n].cpu().numpy()


@pytest.mark.parametrize("name", INDEX_FIXTURES)
def test_search_stages_vs_golden(hip, scorers, name):
    """One query per call; every stage tap compared with the reference's tap."""
    from oracle import oracle as orc
    nat = hip["native"]
    z, scorer = scorers[name]
    oi = orc.OracleIndex.from_golden(z)
    K = int(z["meta.K"])
    for r in rank_records(z):
        pids, scores = _search_one(hip, scorer, z, r, full_table=True)   # the centroid-score tap needs the whole table
        nqc = min(int(z[f"{r}.nq_cand"]), z[f"{r}.Q"].shape[0])
        cs = scorer.tap(nat.TAP_CENTROID_SCORES)[:, :nqc]
        cs_ref = z[f"{r}.centroid_scores"]
        assert np.max(np.abs(cs - cs_ref)) <= GEMM_TOL, r
        # idx bits: identical except where the row max sits within GEMM_TOL of the threshold
        bits = scorer.tap(nat.TAP_IDX_BITS)
        idx = ((bits[np.arange(K) >> 5] >> (np.arange(K) & 31)) & 1).astype(bool)
        near = np.abs(cs_ref.max(-1) - float(z[f"{r}.thr"])) <= GEMM_TOL
        assert np.array_equal(idx[~near], z[f"{r}.idx"][~near]), r
        assert np.array_equal(idx, cs.max(-1) >= np.float32(z[f"{r}.thr"])), r     # self-consistent with its own table
        cells = scorer.tap(nat.TAP_CELLS)
        assert np.array_equal(cells, orc.select_cells(cs, int(z[f"{r}.ncells"]))), r  # exact on its own table
        assert np.array_equal(cells, z[f"{r}.cells"]), r
        cand = scorer.tap(nat.TAP_CANDIDATES)
        assert np.array_equal(cand, z[f"{r}.cand_pids"]), r
        if f"{r}.undefined" in z:
            assert len(set(pids.tolist())) == len(pids) == min(len(cand), int(z[f"{r}.ndocs"]) // 4), r
            continue
        s2 = scorer.tap(nat.TAP_STAGE2)
        # pruning decisions are exact functions of the table bits: compare with the oracle run on THIS table
        assert np.array_equal(s2, oi.filter_pids(cand, cs, idx, int(z[f"{r}.ndocs"]))), r
        assert sorted(s2.tolist()) == sorted(z[f"{r}.filtered_pids"].tolist()), r
        tie_aware_equal(z[f"{r}.final_pids"], z[f"{r}.final_scores"], pids, scores, tol=SCORE_TOL)


@pytest.mark.parametrize("name", INDEX_FIXTURES)
def test_filter_pids_op_bit_exact(hip, name):
    """flmr_filter_pids fed with the reference's own centroid_scores: ids must be bit-exact, in order."""
    torch, ops = hip["torch"], hip["ops"]
    z = load_golden(name)
    doclens = torch.from_numpy(z["index.doclens"])
    offsets = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(doclens, 0)])
    for r in rank_records(z):
        out = ops.filter_pids(torch.from_numpy(z[f"{r}.cand_pids"]), torch.from_numpy(z[f"{r}.centroid_scores"]),
                              torch.from_numpy(z["index.codes"]), doclens, offsets, torch.from_numpy(z[f"{r}.idx"]),
                              int(z[f"{r}.ndocs"]))
        if f"{r}.undefined" in z:
            assert len(set(out.tolist())) == len(out) == min(len(z[f"{r}.cand_pids"]), int(z[f"{r}.ndocs"]) // 4)
        else:
            assert np.array_equal(out.numpy(), z[f"{r}.filtered_pids"]), r


@pytest.mark.parametrize("name", INDEX_FIXTURES)
def test_decompress_op_bit_exact(hip, name):
    torch, ops = hip["torch"], hip["ops"]
    z = load_golden(name)
    doclens = torch.from_numpy(z["index.doclens"])
    offsets = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(doclens, 0)])
    D = ops.decompress_residuals(torch.from_numpy(z["op_decompress.pids"]), doclens, offsets,
                                 torch.from_numpy(z["index.bucket_weights"]), torch.from_numpy(z["codec.reversed_bit_map"]),
                                 torch.from_numpy(z["codec.decompression_lookup_table"]), torch.from_numpy(z["index.residuals"]),
                                 torch.from_numpy(z["index.codes"]), torch.from_numpy(z["index.centroids_f16"].astype(np.float32)),
                                 128, int(z["meta.nbits"]))
    assert np.array_equal(D.numpy().view(np.uint32), z["op_decompress.D"].view(np.uint32))


@pytest.mark.parametrize("name", INDEX_FIXTURES)
def test_lookup_pids_and_eids(hip, scorers, name):
    """IndexScorer.lookup_pids / lookup_eids (index_storage.py:61-65): decompressed + normalised rows of explicit passages /
    tokens; vs the reference's decompress vector normalised with the CPU oracle (fp32, eps 1e-12), bit-exact up to the
    normalisation's last ulp."""
    from oracle import oracle as orc
    torch = hip["torch"]
    z, scorer = scorers[name]
    pids = z["op_decompress.pids"]
    ref = orc.normalize_rows(z["op_decompress.D"])
    D, lens = scorer.lookup_pids(torch.from_numpy(pids))
    assert D.is_cuda and D.dtype == torch.float32 and lens.tolist() == z["index.doclens"][pids].tolist()
    assert D.shape == ref.shape and np.max(np.abs(D.cpu().numpy() - ref)) <= 2e-7
    offsets = np.concatenate([[0], np.cumsum(z["index.doclens"])])
    eids = np.concatenate([np.arange(offsets[p], offsets[p + 1]) for p in pids[:3]])[::-1].copy()
    E = scorer.lookup_eids(torch.from_numpy(eids), out_device="cpu")
    rows = {int(e): i for i, e in enumerate(np.concatenate([np.arange(offsets[p], offsets[p + 1]) for p in pids]))}
    assert np.max(np.abs(E.numpy() - ref[[rows[int(e)] for e in eids]])) <= 2e-7
    # codes override: residual of token e on another centroid (ResidualEmbeddingsStrided.lookup_eids' `codes` argument)
    alt = torch.from_numpy(z["index.codes"][eids]).roll(1)
    E2 = scorer.lookup_eids(torch.from_numpy(eids), codes=alt, out_device="cpu").numpy()
    cen = z["index.centroids_f16"].astype(np.float32)
    raw = z["op_decompress.D"][[rows[int(e)] for e in eids]] - cen[z["index.codes"][eids]] + cen[alt.numpy()]
    assert np.max(np.abs(E2 - orc.normalize_rows(raw.astype(np.float32)))) <= 1e-6


@pytest.mark.parametrize("name", INDEX_FIXTURES)
def test_score_pids_fused_vs_oracle(hip, scorers, name):
    """Fused decompress+normalise+MaxSim (flmr_score_pids) vs the oracle's unfused chain on the same pids."""
    import ctypes as C
    from oracle import oracle as orc
    torch = hip["torch"]
    z, scorer = scorers[name]
    oi = orc.OracleIndex.from_golden(z)
    pids = np.concatenate([z["op_decompress.pids"], np.arange(40, 140, dtype=np.int32)])
    for r in rank_records(z):   # Nq = 28, 32, 48, 96: single-tile kernel and the long-query (LDS-chunked) kernel
        Q = z[f"{r}.Q"]
        ref = orc.maxsim_packed(orc.normalize_rows(oi.decompress(pids)), Q, oi.doclens[pids])
        Qd = torch.from_numpy(Q).cuda()
        pd = torch.from_numpy(pids).cuda()
        # cw: (c.q + w.q) / norm with table-decoded weights (the default for one query tile); regs: decompress-normalise-split
        # wave-per-document kernel (also what long queries take); f32: the fp32-MFMA kernel
        # qs: the query-stationary long-query kernel (the default for Nq > 32; falls to regs for one query tile)
        for impl in ("qs", "lean", "cw", "cwregs", "regs", "f32"):
            with hip["native"].options(FLMR_S3_IMPL=impl):
                out = torch.empty(len(pids), dtype=torch.float32, device="cuda")
                hip["native"].check(scorer._lib.flmr_score_pids(scorer.device_index.handle, C.c_void_p(Qd.data_ptr()), Q.shape[0],
                                                                 C.c_void_p(pd.data_ptr()), len(pids), C.c_void_p(out.data_ptr()),
                                                                 hip["native"].stream_ptr()))
            assert np.max(np.abs(out.cpu().numpy() - ref)) <= SCORE_TOL / 4, (r, impl)


def test_s0_kernel_variants_agree(hip, scorers):
    """fp16-split MFMA (default), fp32 MFMA and the plain k-ascending VALU kernel + table post-pass: the score tables agree
    to fp32-roundoff and every derived integer result (idx bits, cells, candidates) is identical."""
    nat = hip["native"]
    for name, rec in (("idx_nb2", "rank0"), ("idx_nb2", "rank9"), ("idx_nb1", "rank_rz")):
        z, scorer = scorers[name]
        taps = {}
        for impl in ("f16", "f32", "valu"):
            with nat.options(FLMR_S0_IMPL=impl):
                _search_one(hip, scorer, z, rec, full_table=True)
                taps[impl] = [scorer.tap(t) for t in (nat.TAP_CENTROID_SCORES, nat.TAP_IDX_BITS, nat.TAP_CELLS, nat.TAP_CANDIDATES)]
        for impl in ("f16", "f32"):
            assert np.max(np.abs(taps[impl][0] - taps["valu"][0])) <= 5e-7, (name, rec, impl)
            for a, b in zip(taps[impl][1:], taps["valu"][1:]):
                assert np.array_equal(a, b), (name, rec, impl)


@pytest.mark.parametrize("K,npass,nq_list,thr", [
    (2048, 6000, (32, 20, 1), 0.45),     # one table slice per workgroup, ragged query lengths
    (32768, 20_000, (32, 32, 7), 0.45),  # several slices, several surviving rows per query
    (4096, 3000, (32, 5, 32), -1.0),     # every centroid survives: the dense epilogue runs for every tile
    (1024, 2000, (32,), 0.3),            # a single query: 15 of the workgroup's 16 query slots repeat it
])
def test_s0_query_stationary_equals_row_stationary(hip, K, npass, nq_list, thr):
    """Stage 0 on the sparse path, query-stationary kernel (default: queries' hi/lo images in registers, centroid tiles through
    LDS by DMA with hand-counted waits; s0_centroid_scores_qs) vs the row-stationary kernel (FLMR_S0_IMPL=f16rs): same MFMA
    sequence per (row, column), so idx bits, cells, candidates and the final ranking must be identical as bits -- also with
    batches that are not a multiple of 16 queries, ragged query lengths and a threshold every centroid passes."""
    nat = hip["native"]
    torch = hip["torch"]
    from ravqa_amd import synth
    from ravqa_amd.scorer import IndexScorer
    corpus = synth.make_corpus(npass, (8, 60), K, 2, seed=61, device="cuda")
    scorer = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=64)
    outs = {}
    for nb in (1, 21):   # 21: two workgroups of query slots, the second partly filled
        Q, _ = synth.make_queries(corpus, nb, 32, seed=4)
        q_lens = torch.tensor([nq_list[j % len(nq_list)] for j in range(nb)], dtype=torch.int32)
        for impl in ("f16rs", None):
            with nat.options(**({"FLMR_S0_IMPL": impl} if impl else {})):
                p, s, c = scorer.search_batch(Q, 16, 2, thr, 64, 32, q_lens=q_lens)
                torch.cuda.synchronize()
                scorer.check()
                # (the stage-1 survivors are a SET: their order follows the key positions, which the queue form of the scatter
                # kernel hands out with atomics)
                taps = [[scorer.tap(t, q) for t in (nat.TAP_IDX_BITS, nat.TAP_CELLS, nat.TAP_CANDIDATES)] + [np.sort(scorer.tap(nat.TAP_STAGE1, q))]
                        for q in range(nb)]
                outs[impl, nb] = (p.cpu().numpy(), s.cpu().numpy(), c.cpu().numpy(), taps)
        a, b = outs["f16rs", nb], outs[None, nb]
        for q in range(nb):
            for x, y in zip(a[3][q], b[3][q]):
                assert np.array_equal(x, y), (nb, q)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2], b[2]), nb


def test_ops_vs_golden(hip):
    torch, ops = hip["torch"], hip["ops"]
    z = load_golden("ops")
    pids = z["lookup.pids"]
    for tag in ["u8", "i32", "i64", "f32", "f16"]:
        out = ops.segmented_lookup(torch.from_numpy(z[f"lookup.{tag}.input"]), torch.from_numpy(pids),
                                   torch.from_numpy(z["lookup.lengths"][pids]), torch.from_numpy(z["lookup.offsets"][pids]))
        assert out.numpy().tobytes() == z[f"lookup.{tag}.output"].tobytes(), tag
    out = ops.segmented_maxsim(torch.from_numpy(z["maxsim.scores"]), torch.from_numpy(z["maxsim.lengths"]))
    assert out[0] == 0.0 and out[2] == 0.0
    assert np.max(np.abs(out.numpy() - z["maxsim.output"])) <= 1e-5
    out = ops.segmented_maxsim(torch.from_numpy(z["maxsim45.scores"]), torch.from_numpy(z["maxsim.lengths"]))
    assert np.max(np.abs(out.numpy() - z["maxsim45.output"])) <= 1e-5
    out = ops.colbert_score_padded(torch.from_numpy(z["padded.Q"]), torch.from_numpy(z["padded.D"]), torch.from_numpy(z["padded.mask"]))
    assert np.max(np.abs(out.cpu().numpy() - z["padded.output"])) <= SCORE_TOL
    assert float(out[3]) == -9999.0 * 32
    out = ops.colbert_score_padded(torch.from_numpy(z["padded_aligned.Q"]), torch.from_numpy(z["padded.D"]), torch.from_numpy(z["padded.mask"]))
    assert np.max(np.abs(out.cpu().numpy() - z["padded_aligned.output"])) <= SCORE_TOL


def test_batched_equals_single_and_remove_zero(hip, scorers):
    """Batch of different queries == the same queries one by one; q_lens path == host-side row removal."""
    torch = hip["torch"]
    z, scorer = scorers["idx_nb2"]
    recs = ["rank0", "rank3"]  # both Nq=32, same config
    Q = torch.stack([torch.from_numpy(z[f"{r}.Q"]) for r in recs])
    p, s, c = scorer.search_batch(Q.repeat(5, 1, 1), 100, 2, 0.45, 1024, 32)
    for i in range(10):
        r = recs[i % 2]
        n = int(c[i])
        assert n == 100
        tie_aware_equal(z[f"{r}.final_pids"][:100], z[f"{r}.final_scores"][:100], p[i].cpu().numpy(), s[i].cpu().numpy(),
                        tol=SCORE_TOL)
    # remove_zero_tensors: compact on the host, pass q_lens
    from ravqa_amd.searcher import Searcher
    Qraw = torch.from_numpy(z["rank_rz.Q_raw"]).unsqueeze(0)
    Qc, lens = Searcher._compact_nonzero_rows(Qraw)
    assert int(lens[0]) == z["rank_rz.Q"].shape[0]
    assert torch.equal(Qc[0, : int(lens[0])], torch.from_numpy(z["rank_rz.Q"]))
    nd = int(z["rank_rz.ndocs"])
    p, s, c = scorer.search_batch(Qc, nd // 4, int(z["rank_rz.ncells"]), float(z["rank_rz.thr"]), nd, 32, q_lens=lens)
    n = int(c[0])
    tie_aware_equal(z["rank_rz.final_pids"], z["rank_rz.final_scores"], p[0, :n].cpu().numpy(), s[0, :n].cpu().numpy(), tol=SCORE_TOL)


def test_searcher_dropin_api(hip, tmp_path):
    """Searcher(index=..., config=...) under Run().context, _search_all_Q -> Ranking, dense_search, filter_fn path."""
    torch, pkg = hip["torch"], hip["pkg"]
    from ravqa_amd import ColBERTConfig, Queries, Run, RunConfig
    from ravqa_amd.searcher import Searcher
    z = load_golden("idx_nb2")
    root = str(tmp_path)
    pkg.IndexArrays.from_golden(z).save(os.path.join(root, "exp0", "indexes", "temp_index.nbits=2"))
    with Run().context(RunConfig(nranks=1, rank=0, root=root, experiment="exp0")):
        searcher = Searcher(index="temp_index.nbits=2", config=ColBERTConfig(total_visible_gpus=0))
        recs = ["rank0", "rank3"]
        Q = torch.stack([torch.from_numpy(z[f"{r}.Q"]) for r in recs])
        ranking = searcher._search_all_Q(Queries(data={"qa": "what", "qb": "which"}), Q, k=100, progress=False)
        d = ranking.todict()
        assert list(d) == ["qa", "qb"] and searcher.config.ndocs == 1024 and searcher.config.ncells == 2
        for qid, r in zip(d, recs):
            pids, ranks, scores = zip(*d[qid])
            assert list(ranks) == list(range(1, 101))
            tie_aware_equal(z[f"{r}.final_pids"][:100], z[f"{r}.final_scores"][:100], pids, scores, tol=SCORE_TOL)
        pids, ranks, scores = searcher.dense_search(Q[:1], k=10)
        assert pids == list(d["qa"][i][0] for i in range(10)) and ranks == list(range(1, 11))
        # pipelined (default: lists that wait for their device sub-batch) == synchronous, over several sub-batches and every way of
        # reading a list (prefix slice before anything is built, element, iteration, arrays)
        Q9 = torch.cat([Q] * 5)[:9]
        q9 = Queries(data={f"q{i}": "x" for i in range(9)})
        sync = Searcher(index="temp_index.nbits=2", config=ColBERTConfig(total_visible_gpus=0), pipelined=False, max_batch=4)
        pipe = Searcher(index="temp_index.nbits=2", config=ColBERTConfig(total_visible_gpus=0), pipelined=True, max_batch=4)
        ds, dp = sync._search_all_Q(q9, Q9, k=100).todict(), pipe._search_all_Q(q9, Q9, k=100).todict()
        assert dp["q8"][:3] == ds["q8"][:3] and dp["q5"][7] == ds["q5"][7] and len(dp["q4"]) == len(ds["q4"])
        assert dp["q1"].pids.tolist() == ds["q1"].pids.tolist()
        for qid in ds:
            assert list(dp[qid]) == list(ds[qid]) and dp[qid] == ds[qid]
        assert pipe._search_all_Q(q9, Q9, k=100).tolist() == sync._search_all_Q(q9, Q9, k=100).tolist()
        # filter_fn: keep even pids only -> every result even, and equals the oracle on the filtered candidate list
        pids_f, _, scores_f = searcher.dense_search(Q[:1], k=10, filter_fn=lambda p: p[p % 2 == 0])
        assert all(p % 2 == 0 for p in pids_f) and len(pids_f) == 10
        from oracle import oracle as orc
        oi = orc.OracleIndex.from_golden(z)
        cand = z["rank0.cand_pids"]
        cand = cand[cand % 2 == 0]
        cs = z["rank0.centroid_scores"]
        fin = oi.filter_pids(cand, cs, z["rank0.idx"], 1024)
        ref = orc.maxsim_packed(orc.normalize_rows(oi.decompress(fin)), z["rank0.Q"], oi.doclens[fin])
        order = np.argsort(-ref, kind="stable")[:10]
        tie_aware_equal(fin[order], ref[order], pids_f, scores_f, tol=SCORE_TOL)


def test_search_reference_written_index(hip):
    """An index directory written by the reference's own indexer code (two chunks, tests/golden/refindex_2chunk, and
    its legacy-ivf / fp16-table variant) is loaded from disk and searched; expected = the reference's rank() on it."""
    from conftest import GOLDEN
    torch = hip["torch"]
    exp = dict(np.load(os.path.join(GOLDEN, "refindex.npz")))
    Q = torch.from_numpy(exp["rank.Q"]).unsqueeze(0)
    nd = int(exp["rank.ndocs"])
    scorer = hip["IndexScorer"](os.path.join(GOLDEN, "refindex_2chunk"), False)
    p, s, c = scorer.search_batch(Q, nd // 4, int(exp["rank.ncells"]), float(exp["rank.thr"]), nd, 32)
    n = int(c[0])
    tie_aware_equal(exp["rank.pids"], exp["rank.scores"], p[0, :n].cpu().numpy(), s[0, :n].cpu().numpy(), tol=SCORE_TOL)
    # legacy variant: same codes / residuals, bucket weights rounded through fp16 -> same ids, scores within the codec's step
    scorer2 = hip["IndexScorer"](os.path.join(GOLDEN, "refindex_legacy"), False)
    p2, s2, c2 = scorer2.search_batch(Q, nd // 4, int(exp["rank.ncells"]), float(exp["rank.thr"]), nd, 32)
    assert int(c2[0]) == n and sorted(p2[0, :n].tolist()) == sorted(p[0, :n].tolist())
    assert float((s2[0, :n].sort().values - s[0, :n].sort().values).abs().max()) < 5e-3


def test_installed_searcher_through_patched_names(hip, tmp_path):
    """Level-1 drop-in on the device: `ravqa_amd.install()` over a `colbert` package tree (the stand-in of
    tests/fake_colbert.py -- the reference checkout does not exist on the GPU box), then the executor's call sequence
    (FLMR_executor.py:774-794) through the PATCHED names `colbert.Searcher` / `colbert.search.index_storage.IndexScorer`."""
    import sys
    import fake_colbert
    torch, pkg = hip["torch"], hip["pkg"]
    cleanup = fake_colbert.make(str(tmp_path / "pkg"))
    try:
        pkg.install(require_device=True)
        from colbert import Searcher
        from colbert.data import Queries
        from colbert.infra import ColBERTConfig, Run, RunConfig
        from colbert.search.index_storage import IndexScorer
        assert IndexScorer is pkg.IndexScorer and Searcher is pkg.installed()
        z = load_golden("idx_nb2")
        root = str(tmp_path / "ckpt")
        pkg.IndexArrays.from_golden(z).save(os.path.join(root, "temp_index_0", "indexes", "temp_index.nbits=2"))
        with Run().context(RunConfig(nranks=1, rank=0, root=root, experiment="temp_index_0")):
            searcher = Searcher(index="temp_index.nbits=2", config=ColBERTConfig(total_visible_gpus=0))
            assert isinstance(searcher.ranker, pkg.IndexScorer)
            recs = ["rank0", "rank3"]                       # the k <= 100 policy: ncells=2, thr=0.45, ndocs=1024
            Q = torch.stack([torch.from_numpy(z[f"{r}.Q"]) for r in recs])
            ranking = searcher._search_all_Q(Queries(data={i: f"q{i}" for i in range(len(recs))}), Q, k=100)
            for qid, r in enumerate(recs):
                pids, ranks, scores = zip(*ranking.todict()[qid])
                assert list(ranks) == list(range(1, 101))
                tie_aware_equal(z[f"{r}.final_pids"][:100], z[f"{r}.final_scores"][:100], pids, scores, tol=SCORE_TOL)
            # the mid-level name: rank() of the patched IndexScorer == the Searcher's own result
            scorer = IndexScorer(searcher.index, False)
            p0, s0 = scorer.rank(searcher.config, Q[:1])
            assert p0[:100] == [t[0] for t in ranking.todict()[0]]
            # the scoring head through the patched name: ColBERT.score -> colbert_score -> flmr_colbert_score_padded
            # (the call shape of FLMR_executor.py:826-833: Q repeated per item, CUDA tensors, grad off) vs the reference's
            # golden output; with autograd on it must stay with the package's own torch expression
            import colbert.modeling.colbert as mc
            zo = load_golden("ops")
            Qp, Dp, Mp = (torch.from_numpy(zo[f"padded.{n}"]).cuda() for n in ("Q", "D", "mask"))
            model = mc.ColBERT()
            with torch.no_grad():
                got = model.score(Qp, Dp, Mp)
            assert got.is_cuda and got.dtype == torch.float32
            ref = zo["padded.output"]
            assert np.max(np.abs(got.cpu().numpy() - ref) / (1.0 + np.abs(ref))) <= 2e-6
            with torch.no_grad():
                got_a = model.score(torch.from_numpy(zo["padded_aligned.Q"]).cuda(), Dp, Mp.unsqueeze(-1))
            ref_a = zo["padded_aligned.output"]
            assert np.max(np.abs(got_a.cpu().numpy() - ref_a) / (1.0 + np.abs(ref_a))) <= 2e-6
            with hip["native"].options(FLMR_SCORE_IMPL="valu"):      # the patched name really runs the library: a switch of the
                with torch.no_grad():                                 # library changes which kernel answers, not the answer
                    got_v = model.score(Qp, Dp, Mp)
            assert np.max(np.abs(got_v.cpu().numpy() - ref) / (1.0 + np.abs(ref))) <= 2e-6
            Qg = Qp.clone().requires_grad_(True)
            out_g = model.score(Qg, Dp, Mp)
            assert out_g.requires_grad and float((out_g.detach() - got).abs().max()) <= 1e-3
    finally:
        pkg.uninstall()
        cleanup()


@pytest.mark.parametrize("nbits,doclen,nq,nq_cand,K,policy", [
    (2, (1, 200), 32, 32, 2048, (2, 0.45, 256)),
    (4, (100, 300), 96, 48, 2048, (2, 0.45, 256)),      # two column tiles -> full table + table-gather stage 2
    (1, (1, 40), 20, 32, 2048, (2, 0.45, 256)),         # fewer query tokens than the candidate-generation window
    (8, 64, 32, 32, 2048, (2, 0.45, 256)),
    (2, (200, 420), 32, 32, 1024, (1, 0.5, 64)),        # documents longer than 256 tokens (on-demand code loads), ncells=1
    (2, (1, 60), 32, 32, 1000, (2, 0.45, 256)),         # K not a multiple of 64 -> fp32-MFMA S0 + table path
    (2, 32, 64, 32, 4096, (4, 0.4, 4096)),              # the k > 100 policy of searcher.py:108-118
    (2, (8, 64), 32, 32, 2048, (8, 0.4, 1024)),         # ncells = 8: the widest per-token cell list the build supports
    (2, (8, 64), 40, 32, 2048, (3, 0.5, 512)),          # ncells = 3 (runs in the 4-wide instantiation)
])
def test_random_corpus_vs_oracle(hip, nbits, doclen, nq, nq_cand, K, policy):
    """Seeded synthetic corpora at sizes the oracle finishes in seconds: ragged / long docs, Nq != 32, two column tiles,
    odd K, both k-policies."""
    from oracle import oracle as orc
    torch = hip["torch"]
    from ravqa_amd import synth
    from ravqa_amd.scorer import IndexScorer
    corpus = synth.make_corpus(12000 if policy[2] > 1024 else 6000, doclen, K, nbits, seed=5 + nbits, device="cuda")
    Q, _ = synth.make_queries(corpus, 12, nq, seed=9)
    arrays = synth.corpus_to_arrays(corpus)
    scorer = IndexScorer(arrays=arrays)
    oi = orc.OracleIndex(arrays.dim, arrays.nbits, arrays.codes, arrays.residuals, arrays.doclens, arrays.ivf,
                         arrays.ivf_lengths, arrays.centroids, arrays.bucket_weights)
    ncells, thr, ndocs = policy
    p, s, c = scorer.search_batch(Q, ndocs // 4, ncells, thr, ndocs, nq_cand)
    Qh = Q.cpu().numpy()
    for i in range(Q.size(0)):
        rp, rs, ncand = oi.rank(Qh[i], ncells, thr, ndocs, nq_cand)
        n = int(c[i])
        if ncand < ndocs:
            assert n == min(ncand, ndocs // 4)
            continue
        tie_aware_equal(rp, rs, p[i, :n].cpu().numpy(), s[i, :n].cpu().numpy(), tol=SCORE_TOL)


def test_full_size_properties(hip):
    """Size-independent properties at a corpus the oracle cannot sweep: scores sorted, pids unique and in range,
    planted passage retrieved, and the fused scorer is idempotent (re-scoring the returned pids reproduces the scores)."""
    import ctypes as C
    torch = hip["torch"]
    from ravqa_amd import synth
    from ravqa_amd.scorer import IndexScorer
    corpus = synth.make_corpus(200_000, 128, 32768, 2, seed=0, device="cuda")
    scorer = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=64)
    Q, targets = synth.make_queries(corpus, 64, 32)
    p, s, c = scorer.search_batch(Q, 100, 2, 0.45, 1024, 32)
    torch.cuda.synchronize()
    assert int(c.min()) == 100
    assert bool((s[:, :-1] >= s[:, 1:]).all())
    assert int(p.min()) >= 0 and int(p.max()) < 200_000
    for i in range(64):
        assert len(set(p[i].tolist())) == 100
    assert float((p[:, :5] == targets.unsqueeze(1).to(torch.int32)).any(dim=1).float().mean()) >= 0.95
    out = torch.empty(100, dtype=torch.float32, device="cuda")
    for i in (0, 17, 63):
        hip["native"].check(scorer._lib.flmr_score_pids(scorer.device_index.handle, C.c_void_p(Q[i].data_ptr()), 32,
                                                         C.c_void_p(p[i].data_ptr()), 100, C.c_void_p(out.data_ptr()),
                                                         hip["native"].stream_ptr()))
        assert torch.equal(out, s[i])
    # every stage-2 form gives the same bits at this size too (default here: the XCD-sliced kernel; K = 32768 -> 8 MB table)
    ref = (p.cpu(), s.cpu(), c.cpu())
    for impl in ("lds", "xcd", "walk"):
        with hip["native"].options(FLMR_S2_IMPL=impl):
            p2, s2, c2 = scorer.search_batch(Q, 100, 2, 0.45, 1024, 32)
            scorer.check()
        assert torch.equal(p2.cpu(), ref[0]) and torch.equal(s2.cpu().view(torch.int32), ref[1].view(torch.int32)) and torch.equal(c2.cpu(), ref[2]), impl


def test_merge_topk(hip):
    torch, ops = hip["torch"], hip["ops"]
    g = torch.Generator().manual_seed(0)
    sc = torch.rand(4, 7, 10, generator=g).sort(dim=-1, descending=True).values
    pd = torch.stack([torch.stack([torch.randperm(1000, generator=g)[:10] + 1000 * r for _ in range(7)]) for r in range(4)]).int()
    pd[2, 3, 6:] = -1
    ms, mp, mc = ops.merge_topk(sc.cuda(), pd.cuda())
    for q in range(7):
        keys = [(float(sc[r, q, i]), int(pd[r, q, i])) for r in range(4) for i in range(10) if pd[r, q, i] >= 0]
        keys.sort(reverse=True)
        assert mp[q].tolist() == [p for _, p in keys[:10]]
        assert int(mc[q]) == 10


def test_candidate_generation_variants_agree(hip, scorers):
    """Chunked LDS-bitmap candidate generation (default; hit flags fused) vs the first implementation (global atomicOr bitmap,
    separate hit bitmap) vs no hit prefilter at all: identical candidates, stage-1 survivors sets and final results."""
    nat = hip["native"]
    z, scorer = scorers["idx_nb2"]
    outs = {}
    for tag, env in (("chunked", {}), ("atomic", {"FLMR_CAND_IMPL": "atomic"}), ("nohit", {"FLMR_S1_NO_HITMAP": "1"})):
        with nat.options(**env):
            _search_one(hip, scorer, z, "rank0")   # (another query first: no variant may live off the previous one's workspace)
            pids, scores = _search_one(hip, scorer, z, "rank3")
            outs[tag] = (scorer.tap(nat.TAP_CANDIDATES), np.sort(scorer.tap(nat.TAP_STAGE1)), scorer.tap(nat.TAP_STAGE2), pids, scores)
    for tag in ("atomic", "nohit"):
        for a, b in zip(outs["chunked"], outs[tag]):
            assert np.array_equal(a, b), tag


@pytest.mark.parametrize("name", INDEX_FIXTURES)
def test_sparse_table_path_equals_full_table_path(hip, scorers, name):
    """Default path (score table kept only for surviving centroids; stage 2 and the cell probe recompute their scores with
    the S0 MFMA sequence) vs the full-table path: every integer result and every score must be IDENTICAL, bit for bit."""
    nat = hip["native"]
    z, scorer = scorers[name]
    for r in rank_records(z):
        res = {}
        for mode in (False, True):
            pids, scores = _search_one(hip, scorer, z, r, full_table=mode)
            res[mode] = [scorer.tap(t) for t in (nat.TAP_IDX_BITS, nat.TAP_CELLS, nat.TAP_CANDIDATES, nat.TAP_STAGE2)] + [pids, scores]
            res[mode].insert(3, np.sort(scorer.tap(nat.TAP_STAGE1)))
        for a, b in zip(res[False], res[True]):
            assert np.array_equal(a, b), r


def test_build_index_on_gpu_then_search(hip, tmp_path):
    """End to end on the device without the reference or FAISS: embeddings -> indexing.build_index (k-means + compress + IVF,
    torch/rocBLAS ops) -> reference on-disk format -> reload -> HIP search: planted passages come back in the top 5, and the
    HIP result equals the CPU oracle on the same (re-loaded) index."""
    from oracle import oracle as orc
    torch, pkg = hip["torch"], hip["pkg"]
    from ravqa_amd import indexing, synth
    from ravqa_amd.scorer import IndexScorer
    corpus = synth.make_corpus(4000, (8, 40), 512, 2, seed=17, device="cuda")  # used only as an embedding source
    g = torch.Generator(device="cuda").manual_seed(3)
    embs = torch.nn.functional.normalize(corpus.centroids[corpus.codes.long()] + 0.05 * torch.randn(corpus.codes.numel(), 128, generator=g, device="cuda"), dim=-1)
    arrays = indexing.build_index(embs, corpus.doclens, nbits=4, kmeans_niters=4)
    arrays.save(str(tmp_path / "built"))
    re = pkg.load_index_arrays(str(tmp_path / "built"))
    scorer = IndexScorer(arrays=re)
    offs = corpus.doc_offsets
    targets = torch.arange(0, 4000, 125, device="cuda")
    Q = torch.stack([torch.nn.functional.normalize(embs[offs[t]:offs[t + 1]][torch.arange(32, device="cuda") % int(corpus.doclens[t])] +
                                                   0.02 * torch.randn(32, 128, generator=g, device="cuda"), dim=-1) for t in targets])
    p, s, c = scorer.search_batch(Q, 10, 2, 0.45, 256, 32)
    assert float((p[:, :5] == targets.unsqueeze(1).to(torch.int32)).any(dim=1).float().mean()) >= 0.9
    oi = orc.OracleIndex(re.dim, re.nbits, re.codes, re.residuals, re.doclens, re.ivf, re.ivf_lengths, re.centroids, re.bucket_weights)
    Qh = Q.cpu().numpy()
    for i in range(0, len(targets), 4):
        rp, rs, ncand = oi.rank(Qh[i], 2, 0.45, 256)
        if ncand >= 256:
            tie_aware_equal(rp[:10], rs[:10], p[i, : int(c[i])].cpu().numpy(), s[i, : int(c[i])].cpu().numpy(), tol=SCORE_TOL)


def test_colbert_score_padded_mfma_vs_oracle(hip):
    """Padded scoring head (colbert.py:235-286) on the fp16-split MFMA kernel vs the CPU oracle and vs the plain-FMA HIP
    kernel: ragged masks, a fully masked document, Ld not a multiple of 32, Nq spanning two LDS chunks."""
    from oracle import oracle as orc
    torch, ops = hip["torch"], hip["ops"]
    g = torch.Generator().manual_seed(5)
    B, Ld, Nq = 37, 70, 200
    D = torch.nn.functional.normalize(torch.randn(B, Ld, 128, generator=g), dim=-1)
    lens = torch.randint(1, Ld + 1, (B,), generator=g)
    lens[4] = 0
    mask = torch.arange(Ld).unsqueeze(0) < lens.unsqueeze(1)
    for Q in (torch.nn.functional.normalize(torch.randn(1, Nq, 128, generator=g), dim=-1),
              torch.nn.functional.normalize(torch.randn(B, 45, 128, generator=g), dim=-1)):
        ref = orc.colbert_score_padded(Q.numpy(), D.numpy(), mask.numpy())
        got = ops.colbert_score_padded(Q, D, mask).cpu().numpy()
        assert np.max(np.abs(got - ref) / (1.0 + np.abs(ref))) <= 2e-6
        with hip["native"].options(FLMR_SCORE_IMPL="valu"):
            got2 = ops.colbert_score_padded(Q, D, mask).cpu().numpy()
        assert np.max(np.abs(got2 - ref) / (1.0 + np.abs(ref))) <= 2e-6
        assert got[4] == np.float32(-9999.0) * Q.shape[1] or abs(got[4] + 9999.0 * Q.shape[1]) < 1.0


@pytest.mark.parametrize("nshards", [2, 3])
def test_exact_sharded_protocol_equals_unsharded(hip, nshards):
    """SURVEY 8e exact-parity mode: passage shards + three key exchanges reproduce the single-index result bit for bit.
    The ranks are emulated in one process (one IndexScorer per shard on the same GPU, the all-gather is a torch.stack)."""
    torch, pkg, ops = hip["torch"], hip["pkg"], hip["ops"]
    from ravqa_amd.scorer import IndexScorer
    z = load_golden("idx_nb2")
    full = pkg.IndexArrays.from_golden(z)
    single = IndexScorer(arrays=full)
    shards = [IndexScorer(arrays=full.shard(r, nshards)) for r in range(nshards)]
    recs = ["rank0", "rank3"]
    Q = torch.stack([torch.from_numpy(z[f"{r}.Q"]) for r in recs]).repeat(3, 1, 1)
    for (k, ncells, thr, ndocs) in [(100, 2, 0.45, 1024), (10, 1, 0.5, 64)]:
        p_ref, s_ref, c_ref = single.search_batch(Q, k, ncells, thr, ndocs, 32)

        def exchange(keys_per_rank, n):
            g = torch.stack(keys_per_rank)                              # [W, B, m]
            return ops.topn_keys(g.permute(1, 0, 2).reshape(g.size(1), -1), n)

        s1 = exchange([sh.phase1(Q, k, ncells, thr, ndocs, 32) for sh in shards], ndocs)
        s2 = exchange([sh.phase2(s1) for sh in shards], ndocs // 4)
        fin = exchange([sh.phase3(s2) for sh in shards], min(k, ndocs // 4))
        p, s, c = ops.unpack_keys(fin, k)
        assert torch.equal(c, c_ref) and torch.equal(p, p_ref) and torch.equal(s, s_ref), (nshards, k)


@pytest.mark.gpu
@pytest.mark.parametrize("nshards", [2, 3])
def test_query_split_stage0_equals_unsharded(hip, nshards):
    """Phase 0 of the sharded protocol: each rank runs stage 0 for a slice of the queries only (flmr_search_probe), the
    (idx bitset, cells, ncell) triples are gathered and flmr_search_phase1_probed rebuilds the table rows -- the final
    ranking must still be bit-identical to the unsharded search."""
    torch, pkg, ops = hip["torch"], hip["pkg"], hip["ops"]
    from ravqa_amd.scorer import IndexScorer
    z = load_golden("idx_nb2")
    full = pkg.IndexArrays.from_golden(z)
    single = IndexScorer(arrays=full)
    shards = [IndexScorer(arrays=full.shard(r, nshards)) for r in range(nshards)]
    recs = ["rank0", "rank3"]
    Q = torch.stack([torch.from_numpy(z[f"{r}.Q"]) for r in recs]).repeat(4, 1, 1)[:7]   # 7 queries: ragged slices
    B = Q.size(0)
    per = -(-B // nshards)
    q_lens = torch.tensor([32, 17, 32, 1, 32, 32, 25], dtype=torch.int32)
    # thr=-1: every centroid qualifies; ndocs=4096 > #passages: every exchanged list is mostly empty keys
    for (k, ncells, thr, ndocs) in [(100, 2, 0.45, 1024), (10, 1, 0.5, 64), (10, 2, -1.0, 64), (200, 4, 0.4, 4096)]:
        p_ref, s_ref, c_ref = single.search_batch(Q, k, ncells, thr, ndocs, 32, q_lens=q_lens)

        def exchange(keys_per_rank, n, ordered=False):   # the intermediate exchanges use the unordered radix select
            g = torch.stack(keys_per_rank)
            return ops.topn_keys(g.permute(1, 0, 2).reshape(g.size(1), -1), n, ordered=ordered)

        parts = []
        for r, sh in enumerate(shards):
            lo = min(B, r * per)
            parts.append(sh.probe(Q, k, ncells, thr, ndocs, lo, min(B, lo + per) - lo, 32, q_lens=q_lens))
        bits, cells, ncell = (torch.cat([p_[j] for p_ in parts]) for j in range(3))
        # the probe state must equal what the unsharded search computed
        for q in range(B):
            single.search_batch(Q[q:q + 1], k, ncells, thr, ndocs, 32, q_lens=q_lens[q:q + 1])
            ref_bits = single.tap(pkg._native.TAP_IDX_BITS, 0).view(np.int32)
            assert np.array_equal(bits[q].cpu().numpy(), ref_bits), q
            ref_cells = single.tap(pkg._native.TAP_CELLS, 0)
            assert np.array_equal(cells[q, :int(ncell[q])].cpu().numpy(), ref_cells), q
        keys1 = [sh.phase1_probed(Q, k, ncells, thr, ndocs, bits, cells, ncell, 32, q_lens=q_lens) for sh in shards]
        s1 = exchange(keys1, ndocs)
        s1_sorted = exchange(keys1, ndocs, ordered=True)
        # every rank derives this list on its own from the same gathered keys: the select must be reproducible
        for _ in range(3):
            assert torch.equal(s1, exchange(keys1, ndocs))
        # a rank's phase-1 keys are a SET (their order follows the key positions the scatter stage 1 hands out with atomics):
        # running phase 1 again must give the same set
        for _ in range(2):
            again = [sh.phase1_probed(Q, k, ncells, thr, ndocs, bits, cells, ncell, 32, q_lens=q_lens) for sh in shards]
            assert torch.equal(s1_sorted, exchange(again, ndocs, ordered=True))
        u = lambda t: t.cpu().numpy().view(np.uint64)                                   # keys are u64 bit patterns
        assert np.array_equal(np.sort(u(s1), axis=1)[:, ::-1], u(s1_sorted))             # same set as the bitonic top-n
        # phase-2/3 outputs are slot-aligned with the global list: one non-zero contributor per slot -> SUM "all-reduce"
        parts2 = [sh.phase2(s1) for sh in shards]
        assert int((torch.stack(parts2) != 0).sum(dim=0).max()) <= 1
        s2 = ops.topn_keys(torch.stack(parts2).sum(dim=0), ndocs // 4, ordered=False)
        fin = ops.topn_keys(torch.stack([sh.phase3(s2) for sh in shards]).sum(dim=0), min(k, ndocs // 4), ordered=True)
        p, s, c = ops.unpack_keys(fin, k)
        assert torch.equal(c, c_ref) and torch.equal(p, p_ref) and torch.equal(s, s_ref), (nshards, k, thr)


@pytest.mark.parametrize("name", ["idx_nb1", "idx_nb2", "idx_nb4", "idx_nb8"])
def test_compress_ops_match_reference_vectors(hip, name):
    """Index build ops (flmr_nearest_centroids + flmr_compress_residuals) vs the reference's ResidualCodec.compress output
    (tests/golden: compress.embs -> compress.codes / compress.residuals), residual.py:169-222.  Codes may differ only where
    the two best centroids are within fp32 rounding of each other; the packed bytes must be bit-exact for equal codes."""
    torch, ops = hip["torch"], hip["ops"]
    z = load_golden(name)
    nbits = int(z["meta.nbits"])
    cen = torch.from_numpy(z["index.centroids_f16"].astype(np.float32))
    embs = torch.from_numpy(z["compress.embs"])
    codes = ops.nearest_centroids(embs, cen).cpu().numpy()
    ref_codes = z["compress.codes"]
    diff = np.nonzero(codes != ref_codes)[0]
    if diff.size:
        sc = embs.numpy()[diff].astype(np.float64) @ cen.numpy().astype(np.float64).T
        gap = np.abs(sc[np.arange(diff.size), codes[diff]] - sc[np.arange(diff.size), ref_codes[diff]])
        assert gap.max() <= 1e-6, gap.max()
    res = ops.compress_residuals(embs, cen, torch.from_numpy(ref_codes), torch.from_numpy(z["index.bucket_cutoffs"]), nbits)
    assert np.array_equal(res.cpu().numpy(), z["compress.residuals"])


def test_nearest_centroids_large_and_odd_sizes(hip):
    """fp16-split MFMA argmax vs an fp64 argmax: K not a multiple of 64 (padded inside), n not a multiple of 32, several
    workspace chunks, exact ties (duplicate centroids -> the lower index wins)."""
    torch, ops = hip["torch"], hip["ops"]
    from ravqa_amd import synth
    g = torch.Generator().manual_seed(5)
    for (K, n) in [(1000, 4133), (4096, 150_001)]:
        cen = torch.nn.functional.normalize(torch.randn(K, 128, generator=g), dim=-1).half().float()
        cen[K // 2] = cen[3]                                         # exact duplicate: index 3 must win
        codes_true = torch.randint(0, K, (n,), generator=g)
        embs = torch.nn.functional.normalize(cen[codes_true] + 0.05 * torch.randn(n, 128, generator=g), dim=-1)
        got = ops.nearest_centroids(embs, cen).cpu()
        sc = (embs.cuda().double() @ cen.cuda().double().T)
        ref = sc.argmax(dim=1).cpu()
        bad = torch.nonzero(got.long() != ref).flatten()
        if bad.numel():
            gap = (sc[bad.cuda(), ref[bad].cuda()] - sc[bad.cuda(), got[bad].long().cuda()]).abs().max().item()
            assert gap <= 1e-6, gap
        assert not bool((got == K // 2).any())
        # fused residual / bucketize / bit-pack == the torch restatement (bit-exact), every nbits
        for nbits in (1, 2, 4, 8):
            cut, _ = synth.bucket_tables((embs - cen[got.long()])[:20000], nbits)
            want = synth.compress(embs, cen, cut, nbits, codes=got.long())[1]
            have = ops.compress_residuals(embs, cen, got, cut, nbits).cpu()
            assert torch.equal(have, want), (K, n, nbits)


def test_nearest_centroids_rejects_non_fp16_centroids(hip):
    torch, ops, pkg = hip["torch"], hip["ops"], hip["pkg"]
    cen = torch.nn.functional.normalize(torch.randn(128, 128), dim=-1)  # not rounded to half
    with pytest.raises(pkg.FlmrNativeError):
        ops.nearest_centroids(torch.randn(64, 128), cen)


def test_exhaustive_search_matches_torch_expression(hip):
    """FLMR_executor.py:799-847 exhaustive branch: rate_batch from the padded HIP scorer == the torch expression of
    colbert_score (colbert.py:235-286: D @ Q^T, mask-fill -9999, max over Ld, sum over Nq), top-k order identical."""
    torch = hip["torch"]
    from ravqa_amd import scoring
    g = torch.Generator().manual_seed(11)
    nq, Nq, n_items, Ld = 5, 32, 203, 41
    Q = torch.nn.functional.normalize(torch.randn(nq, Nq, 128, generator=g), dim=-1)
    D = torch.nn.functional.normalize(torch.randn(n_items, Ld, 128, generator=g), dim=-1)
    lens = torch.randint(1, Ld + 1, (n_items,), generator=g)
    mask = (torch.arange(Ld).unsqueeze(0) < lens.unsqueeze(1)).unsqueeze(-1)
    idx, sc, rate = scoring.exhaustive_search(Q, D, mask, 10, item_chunk=64)
    ref = torch.empty(nq, n_items, dtype=torch.float64)
    for q in range(nq):
        s = (D.double() @ Q[q].double().T)                          # [n_items, Ld, Nq]
        s = s.masked_fill(~mask.expand(-1, -1, Nq), -9999.0)
        ref[q] = s.max(1).values.sum(-1)
    assert float((rate.cpu().double() - ref).abs().max()) <= 1e-4
    ref_sorted, ref_idx = torch.sort(ref, dim=-1, descending=True)
    for q in range(nq):
        tie_aware_equal(ref_idx[q, :10].numpy(), ref_sorted[q, :10].float().numpy(), idx[q].cpu().numpy(), sc[q].cpu().numpy(), tol=SCORE_TOL)
    rd = scoring.exhaustive_ranking_dict(idx, sc)
    assert rd[0][0] == (int(idx[0, 0]), 0, int(sc[0, 0])) and len(rd) == nq and len(rd[0]) == 10


@pytest.mark.parametrize("nbits,doclen,K,npass,policy", [
    (2, (1, 200), 2048, 6000, (2, 0.45, 256)),
    (2, 64, 512, 70_000, (2, 0.3, 1024)),      # small K on a 3-chunk corpus: > 480 hit candidates per chunk -> several slot windows
    (4, (10, 90), 1024, 40_000, (4, 0.4, 4096)),
    (2, (20, 120), 4096, 20_000, (1, 0.3, 256)),   # one cell per token, low threshold: surviving centroids that are NOT probed cells
    (2, 128, 8192, 50_000, (2, 0.45, 8192)),       # more survivors wanted than there are hit candidates: the misses are appended
])
def test_stage1_scatter_equals_code_scan(hip, nbits, doclen, K, npass, policy):
    """Stage 1 computed from the surviving centroids' IVF lists (cand_mark_score_kernel, default) must give exactly the keys
    of the code-scanning kernel (filter_stage1_kernel, FLMR_S1_IMPL=scan): same stage-1 survivor sets, same stage-2
    finalists, bitwise the same final result.  The scan kernel is itself pinned to the reference by the golden-vector tests."""
    nat, torch = hip["native"], hip["torch"]
    from ravqa_amd import synth
    from ravqa_amd.scorer import IndexScorer
    corpus = synth.make_corpus(npass, doclen, K, nbits, seed=31, device="cuda")
    Q, _ = synth.make_queries(corpus, 9, 32, seed=4)
    q_lens = torch.tensor([32, 32, 20, 32, 1, 32, 0, 7, 32], dtype=torch.int32)   # incl. an empty query (every stage-1 score ties)
    scorer = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=16)
    ncells, thr, ndocs = policy
    outs = {}
    for tag, env in (("scatter", {}), ("scan", {"FLMR_S1_IMPL": "scan"})):
        with nat.options(**env):
            p, s, c = scorer.search_batch(Q, ndocs // 4, ncells, thr, ndocs, 32, q_lens=q_lens)
            torch.cuda.synchronize()
            taps = [(np.sort(scorer.tap(nat.TAP_STAGE1, q)), scorer.tap(nat.TAP_STAGE2, q)) for q in range(Q.size(0))]
            outs[tag] = (p.cpu().numpy(), s.cpu().numpy(), c.cpu().numpy(), taps)
    a, b = outs["scatter"], outs["scan"]
    for q in range(Q.size(0)):
        assert np.array_equal(a[3][q][0], b[3][q][0]), ("stage-1 survivors", q)
        assert np.array_equal(a[3][q][1], b[3][q][1]), ("stage-2 finalists", q)
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


@pytest.mark.parametrize("thr,ncells,ndocs", [(0.99, 2, 64), (0.99, 1, 256), (-1.0, 2, 64)])
def test_degenerate_thresholds_and_empty_passages_vs_oracle(hip, thr, ncells, ndocs):
    """Edge cases of the pruning stages against the CPU oracle: a threshold no centroid reaches (every candidate gets the
    all-miss stage-1 score, so the top-ndocs cut is decided by the (score, pid) tie order alone -- the priority-queue order
    of filter_pids.cpp:24), a threshold every centroid passes (hit_valid = 0: the code-scanning stage 1 runs), and a corpus
    that contains EMPTY passages (doclen 0) and one-token passages."""
    from oracle import oracle as orc
    torch = hip["torch"]
    from ravqa_amd import synth
    from ravqa_amd.scorer import IndexScorer
    corpus = synth.make_corpus(3000, (0, 12), 256, 2, seed=91, device="cuda")
    assert int((corpus.doclens == 0).sum()) > 50
    Q, _ = synth.make_queries(corpus, 6, 32, seed=3)
    arrays = synth.corpus_to_arrays(corpus)
    assert arrays.check_ivf_invariant()
    scorer = IndexScorer(arrays=arrays)
    oi = orc.OracleIndex(arrays.dim, arrays.nbits, arrays.codes, arrays.residuals, arrays.doclens, arrays.ivf,
                         arrays.ivf_lengths, arrays.centroids, arrays.bucket_weights)
    p, s, c = scorer.search_batch(Q, ndocs // 4, ncells, thr, ndocs, 32)
    Qh = Q.cpu().numpy()
    checked = 0
    for i in range(Q.size(0)):
        rp, rs, ncand = oi.rank(Qh[i], ncells, thr, ndocs, 32)
        n = int(c[i])
        if ncand < ndocs:   # the reference's behaviour is undefined there; this build keeps every candidate
            assert n == min(ncand, ndocs // 4)
            continue
        tie_aware_equal(rp, rs, p[i, :n].cpu().numpy(), s[i, :n].cpu().numpy(), tol=SCORE_TOL)
        checked += 1
    assert checked >= 1


def test_key_selection_kernels_on_crafted_rows(hip):
    """flmr_select_keys / flmr_topn_keys (radix select with register-resident keys, LDS bucket compaction, early exit,
    scan-placed output; bitonic sort) against numpy on rows built to hit every branch: fewer non-empty keys than n, many
    equal high bytes (deep passes), rows longer than the register window (streamed tail), m not a multiple of the block size,
    and reproducibility of the unordered form."""
    torch, ops = hip["torch"], hip["ops"]
    rng = np.random.default_rng(7)
    cases = []
    for (m, n) in [(700, 256), (1024, 1024), (8192, 1024), (30011, 1024), (8192, 8), (5000, 4096)]:
        rows = []
        # distinct random keys; keys sharing their top 5 bytes (forces the last passes); a row with only 37 non-empty keys
        rows.append(rng.integers(1, 2**63, size=m, dtype=np.uint64) * np.uint64(2) + np.uint64(1))
        base = np.uint64(0xC1F3A2B4C5000000)
        rows.append(base + rng.permutation(m).astype(np.uint64))
        sparse = np.zeros(m, dtype=np.uint64)
        sparse[rng.choice(m, size=min(37, m), replace=False)] = rng.integers(1, 2**62, size=min(37, m), dtype=np.uint64)
        rows.append(sparse)
        cases.append((m, n, np.stack(rows)))
    for m, n, keys in cases:
        t = torch.from_numpy(keys.view(np.int64)).cuda()
        want = -np.sort(-keys.astype(np.float64), axis=1)  # only used for shape; exact compare below on uint64
        ref = np.sort(keys, axis=1)[:, ::-1][:, :n]
        got_sorted = ops.topn_keys(t, n, ordered=True).cpu().numpy().view(np.uint64)
        assert np.array_equal(got_sorted, ref), (m, n, "sorted")
        a = ops.topn_keys(t, n, ordered=False)
        b = ops.topn_keys(t, n, ordered=False)
        assert torch.equal(a, b), (m, n, "reproducible")
        got_set = np.sort(a.cpu().numpy().view(np.uint64), axis=1)[:, ::-1]
        assert np.array_equal(got_set, ref), (m, n, "unordered set")


def test_c_abi_reports_errors_instead_of_crashing(hip, scorers):
    """Boundary behaviour (status codes, no exceptions across the C ABI, thread-local message): out-of-range
    parameters, NULL pointers and capacity violations return the documented status with a message; nothing is launched."""
    import ctypes as C
    torch, nat = hip["torch"], hip["native"]
    lib = nat.load()
    z, scorer = scorers["idx_nb2"]
    ix = scorer.device_index.handle
    OK, INVALID, UNSUPPORTED, CAPACITY = 0, 1, 2, 5

    def params(k=10, ncells=2, thr=0.45, ndocs=64, nq_cand=32):
        return nat.SearchParams(k, ncells, thr, ndocs, nq_cand)

    h = C.c_void_p()
    for bad, want in ((params(ncells=9), UNSUPPORTED), (params(ncells=0), UNSUPPORTED), (params(ndocs=2), UNSUPPORTED),
                      (params(ndocs=1 << 20), UNSUPPORTED), (params(nq_cand=0), UNSUPPORTED), (params(k=0), INVALID)):
        assert lib.flmr_searcher_create(ix, 4, 32, C.byref(bad), C.byref(h)) == want
        assert len(lib.flmr_last_error()) > 0 and not h.value
    assert lib.flmr_searcher_create(None, 4, 32, C.byref(params()), C.byref(h)) == INVALID
    assert lib.flmr_searcher_create(ix, 0, 32, C.byref(params()), C.byref(h)) == INVALID
    assert lib.flmr_searcher_create(ix, 4, 32, C.byref(params()), C.byref(h)) == OK and h.value
    try:
        Q = torch.randn(8, 32, 128, device="cuda")
        out_p = torch.empty((8, 10), dtype=torch.int32, device="cuda")
        out_s = torch.empty((8, 10), dtype=torch.float32, device="cuda")
        out_c = torch.empty((8,), dtype=torch.int32, device="cuda")
        ptr = lambda t: C.c_void_p(t.data_ptr())
        st = nat.stream_ptr()
        p = params()
        # more queries / longer queries / wider policy than the workspace was created for
        assert lib.flmr_search_batch(h, ptr(Q), None, 8, 32, C.byref(p), ptr(out_p), ptr(out_s), ptr(out_c), st) == CAPACITY
        assert lib.flmr_search_batch(h, ptr(Q), None, 4, 33, C.byref(p), ptr(out_p), ptr(out_s), ptr(out_c), st) == CAPACITY
        assert lib.flmr_search_batch(h, ptr(Q), None, 4, 32, C.byref(params(ndocs=128)), ptr(out_p), ptr(out_s), ptr(out_c), st) == CAPACITY
        assert lib.flmr_search_batch(h, ptr(Q), None, 4, 32, C.byref(params(ncells=4)), ptr(out_p), ptr(out_s), ptr(out_c), st) == CAPACITY
        assert lib.flmr_search_batch(h, None, None, 4, 32, C.byref(p), ptr(out_p), ptr(out_s), ptr(out_c), st) == INVALID
        assert lib.flmr_search_batch(h, ptr(Q), None, 4, 32, C.byref(p), None, ptr(out_s), ptr(out_c), st) == INVALID
        assert lib.flmr_search_batch(h, ptr(Q), None, 0, 32, C.byref(p), ptr(out_p), ptr(out_s), ptr(out_c), st) == CAPACITY
        # the phased protocol insists on ndocs == the workspace's ndocs; the probe slice must lie inside the batch
        keys = torch.empty((4, 64), dtype=torch.int64, device="cuda")
        assert lib.flmr_search_phase1(h, ptr(Q), None, 4, 32, C.byref(params(ndocs=32)), ptr(keys), st) == INVALID
        iw, mc = C.c_int32(0), C.c_int32(0)
        assert lib.flmr_searcher_probe_dims(h, C.byref(iw), C.byref(mc)) == OK and iw.value > 0 and mc.value == 64
        bits = torch.empty((4, iw.value), dtype=torch.int32, device="cuda")
        cells = torch.empty((4, mc.value), dtype=torch.int32, device="cuda")
        ncell = torch.empty((4,), dtype=torch.int32, device="cuda")
        assert lib.flmr_search_probe(h, ptr(Q), None, 4, 32, C.byref(p), 2, 3, ptr(bits), ptr(cells), ptr(ncell), st) == INVALID
        # and a valid call still works afterwards
        assert lib.flmr_search_batch(h, ptr(Q), None, 4, 32, C.byref(p), ptr(out_p), ptr(out_s), ptr(out_c), st) == OK
        torch.cuda.synchronize()
        # op-level entry points
        assert lib.flmr_select_keys(None, 4, 64, 16, ptr(keys), st) == INVALID
        assert lib.flmr_select_keys(ptr(keys), 4, 0, 16, ptr(keys), st) == INVALID
        assert lib.flmr_compress_residuals(ptr(Q), 128, ptr(Q), ptr(out_c), 8, ptr(out_s), 3, ptr(out_p), st) == UNSUPPORTED
    finally:
        lib.flmr_searcher_destroy(h)


def test_deferred_device_errors_and_q_lens_clamp(hip, scorers):
    """q_lens outside [0, nq] cannot be seen by the host without a sync: the kernels read a clamped copy (so the batch runs
    memory-safe and equals the run with the clamped lengths), a device flag is raised, and the NEXT call on the searcher --
    or flmr_searcher_check at once -- returns FLMR_ERR_INVALID and clears it (include/flmr_hip.h: flmr_searcher_check)."""
    torch, nat = hip["torch"], hip["native"]
    z, scorer = scorers["idx_nb2"]
    Q = torch.stack([torch.from_numpy(z[f"rank{i}.Q"]) for i in (0, 3, 1)])
    good = torch.tensor([32, 20, 32], dtype=torch.int32)
    bad = torch.tensor([77, 20, 1 << 30], dtype=torch.int32)  # both out-of-range entries clamp to nq = 32
    ref = scorer.search_batch(Q, 16, 2, 0.45, 64, 32, q_lens=good)
    scorer.check()
    got = scorer.search_batch(Q, 16, 2, 0.45, 64, 32, q_lens=bad)
    torch.cuda.synchronize()
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    with pytest.raises(nat.FlmrNativeError, match="q_lens"):
        scorer.check()
    scorer.check()                                             # reported once, then clear
    # without an explicit check the error surfaces on the next batch call
    scorer.search_batch(Q, 16, 2, 0.45, 64, 32, q_lens=bad)
    torch.cuda.synchronize()
    with pytest.raises(nat.FlmrNativeError, match="q_lens"):
        scorer.search_batch(Q, 16, 2, 0.45, 64, 32, q_lens=good)
    scorer.search_batch(Q, 16, 2, 0.45, 64, 32, q_lens=good)
    scorer.check()


def test_switches_are_snapshotted_not_read_from_the_environment(hip, scorers):
    """FLMR_* switches: the environment is read once per process; later changes go through flmr_set_option, and a native
    searcher keeps the snapshot it was created with (nothing on the launch path calls getenv)."""
    import ctypes as C
    torch, nat = hip["torch"], hip["native"]
    lib = nat.load()
    z, scorer = scorers["idx_nb2"]
    assert lib.flmr_set_option(b"FLMR_NOT_A_SWITCH", b"1") == 1 and b"unknown option" in lib.flmr_last_error()
    assert lib.flmr_set_option(b"FLMR_S1_IMPL", b"x" * 40) == 1
    Q = torch.from_numpy(z["rank0.Q"]).unsqueeze(0)
    p = nat.SearchParams(8, 2, 0.45, 64, 32)
    ok = C.c_int32(-1)

    def supported():
        h = C.c_void_p()
        nat.check(lib.flmr_searcher_create(scorer.device_index.handle, 1, 32, C.byref(p), C.byref(h)))
        nat.check(lib.flmr_searcher_probe_supported(h, 32, C.byref(p), C.byref(ok)))
        return h, ok.value

    h1, s1 = supported()
    os.environ["FLMR_FULL_TABLE"] = "1"                      # too late: the environment was resolved at first use
    try:
        h2, s2 = supported()
        nat.set_option("FLMR_FULL_TABLE", "1")              # the API does change what NEW searchers see ...
        h3, s3 = supported()
        nat.check(lib.flmr_searcher_probe_supported(h1, 32, C.byref(p), C.byref(ok)))   # ... and old ones keep theirs
        assert (s1, s2, s3, ok.value) == (1, 1, 0, 1)
    finally:
        os.environ.pop("FLMR_FULL_TABLE", None)
        nat.set_option("FLMR_FULL_TABLE", None)
        for h in (h1, h2, h3):
            lib.flmr_searcher_destroy(h)
    # nq_cand > 32 -> two column tiles -> no sparse table -> no query split
    p2 = nat.SearchParams(8, 2, 0.45, 64, 48)
    h = C.c_void_p()
    nat.check(lib.flmr_searcher_create(scorer.device_index.handle, 1, 96, C.byref(p2), C.byref(h)))
    nat.check(lib.flmr_searcher_probe_supported(h, 96, C.byref(p2), C.byref(ok)))
    lib.flmr_searcher_destroy(h)
    assert ok.value == 0


@pytest.mark.parametrize("nbits,doclen,K,npass,policy", [
    (2, (1, 200), 2048, 6000, (2, 0.45, 256)),      # ragged passages incl. one-token ones, fewer survivors than a work item holds
    (2, 64, 512, 70_000, (2, 0.3, 1024)),           # K = one LDS slice: many tokens of a passage fall into the same slice
    (4, (10, 90), 1024, 40_000, (4, 0.4, 4096)),    # ndocs = 4096: four work items per query
    (2, (300, 420), 4096, 3000, (2, 0.45, 256)),    # passages longer than 256 tokens, several code windows per slice
])
def test_stage2_walk_equals_gather(hip, nbits, doclen, K, npass, policy):
    """Stage 2 as the query-stationary table walk (flmr_stage2_walk.hip, FLMR_S2_IMPL=walk: sorted per-passage codes, score
    slices in LDS, running maxima in registers) vs the row-gather kernel (FLMR_S2_IMPL=lds): the stage-2 finalists IN ORDER,
    and the final ids, scores (as bits) and counts must be identical -- max is exact, so the token order cannot matter."""
    nat = hip["native"]
    torch = hip["torch"]
    from ravqa_amd import synth
    from ravqa_amd.scorer import IndexScorer
    corpus = synth.make_corpus(npass, doclen, K, nbits, seed=41, device="cuda")
    Q, _ = synth.make_queries(corpus, 9, 32, seed=6)
    q_lens = torch.tensor([32, 32, 20, 32, 1, 32, 32, 7, 32], dtype=torch.int32)
    scorer = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=16)
    ncells, thr, ndocs = policy
    outs = {}
    for impl in ("lds", "walk"):
        with nat.options(FLMR_S2_IMPL=impl):
            for tag, ql in (("full", None), ("ragged", q_lens)):
                p, s, c = scorer.search_batch(Q, ndocs // 4, ncells, thr, ndocs, 32, q_lens=ql)
                torch.cuda.synchronize()
                taps = [scorer.tap(nat.TAP_STAGE2, q) for q in range(Q.size(0))]
                outs[impl, tag] = (p.cpu().numpy(), s.cpu().numpy(), c.cpu().numpy(), taps)
            scorer.check()
    for tag in ("full", "ragged"):
        a, b = outs["lds", tag], outs["walk", tag]
        for q in range(Q.size(0)):
            assert np.array_equal(a[3][q], b[3][q]), ("stage-2 finalists", tag, q)
        assert np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), tag


@pytest.mark.parametrize("nbits,doclen,npids", [
    (1, (0, 70), 3000), (2, (0, 12), 5000), (2, (1, 300), 777), (4, (30, 34), 4096), (8, (0, 200), 2500), (2, 128, 1),
])
def test_s3_dma_kernel_equals_register_kernel(hip, nbits, doclen, npids):
    """S3's default kernel (centroid rows by LDS-DMA two tiles ahead, hand-counted vmcnt waits; flmr_maxsim.hip) vs the register
    row-gather kernel (FLMR_S3_IMPL=regs): same arithmetic in the same order, so the scores must agree bit for bit -- over
    empty passages, one-token passages, passages of several tiles, tile counts that end the pipeline at every phase, ragged
    query lengths via the search path, and a single document."""
    import ctypes as C
    nat = hip["native"]
    torch = hip["torch"]
    from ravqa_amd import synth
    from ravqa_amd.scorer import IndexScorer
    corpus = synth.make_corpus(6000, doclen, 512, nbits, seed=77 + nbits, device="cuda")
    scorer = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=8)
    Q, _ = synth.make_queries(corpus, 6, 32, seed=9)
    g = torch.Generator().manual_seed(5)
    pids = torch.randperm(6000, generator=g)[:npids].to(torch.int32).cuda()
    outs = {}
    for impl in ("regs", "dma", "cw", "cwregs", "lean"):
        with nat.options(FLMR_S3_IMPL=impl):
            res = []
            for nq in (32, 17):
                Qd = Q[0, :nq].contiguous()
                out = torch.full((npids,), -7.0, dtype=torch.float32, device="cuda")
                nat.check(scorer._lib.flmr_score_pids(scorer.device_index.handle, C.c_void_p(Qd.data_ptr()), nq,
                                                      C.c_void_p(pids.data_ptr()), npids, C.c_void_p(out.data_ptr()),
                                                      nat.stream_ptr()))
                res.append(out.cpu().numpy())
            q_lens = torch.tensor([32, 5, 32, 1, 20, 32], dtype=torch.int32)
            for ql in (None, q_lens):
                p, s, c = scorer.search_batch(Q, 64, 2, 0.3, 256, 32, q_lens=ql)
                scorer.check()
                res += [p.cpu().numpy(), s.cpu().numpy(), c.cpu().numpy()]
            outs[impl] = res
    for a, b in zip(outs["regs"], outs["dma"]):
        assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
    assert not np.any(outs["dma"][0] == -7.0)
    # the centroid + weight form (the default) rounds differently -- (c.q + w.q) * 1/norm instead of ((c + w) / norm).q -- by
    # fp32-roundoff-class terms: same scores to a few 1e-6 at |score| <= 32, same passages except across such near-ties
    assert not np.any(outs["cw"][0] == -7.0)
    for other in ("cwregs", "lean"):   # the pipelines of the same arithmetic agree bit for bit (lean: planned tiles, the default)
        for a, b in zip(outs["cw"], outs[other]):
            assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b), other
    for k in (0, 1):
        assert np.max(np.abs(outs["cw"][k] - outs["regs"][k])) <= 2e-5, (k, np.max(np.abs(outs["cw"][k] - outs["regs"][k])))
    for k in (2, 5):   # search results: pids / scores / counts
        pr, sr, cr = outs["regs"][k:k + 3]
        pc, sc, cc = outs["cw"][k:k + 3]
        assert np.array_equal(cr, cc)
        for q in range(pr.shape[0]):
            n = int(cr[q])
            tie_aware_equal(pr[q, :n], sr[q, :n], pc[q, :n], sc[q, :n], gap=4e-5, tol=2e-5)


@pytest.mark.parametrize("nbits,doclen,K,npass,policy", [
    (2, (0, 200), 2048, 6000, (2, 0.45, 256)),      # ragged passages incl. empty and one-token ones; most (passage, slice) runs < 8 tokens
    (2, 64, 512, 70_000, (2, 0.3, 1024)),           # 64-code slices: runs of several octets, four survivor groups per query
    (4, (10, 90), 1000, 40_000, (4, 0.4, 4096)),    # K not a multiple of 8 slices; ndocs = 4096: sixteen waves per (query, slice)
    (2, (300, 420), 4096, 3000, (2, 0.45, 256)),    # passages longer than 256 tokens
    (2, 128, 32768, 20_000, (2, 0.45, 1024)),       # a table beyond one L2: the kernel is the DEFAULT here (no switch set)
])
def test_stage2_xcd_sliced_equals_gather(hip, nbits, doclen, K, npass, policy):
    """Stage 2 with the centroid table cut into one slice per XCD (flmr_stage2_xcd.hip: sorted per-passage codes, 8-token
    octets, per-slice column maxima + combine) vs the row-gather kernel (FLMR_S2_IMPL=lds): the stage-2 finalists IN ORDER,
    and the final ids, scores (as bits) and counts must be identical -- max is exact, so neither the token order nor the
    cut into slices can matter."""
    nat = hip["native"]
    torch = hip["torch"]
    from ravqa_amd import synth
    from ravqa_amd.scorer import IndexScorer
    corpus = synth.make_corpus(npass, doclen, K, nbits, seed=43, device="cuda")
    Q, _ = synth.make_queries(corpus, 9, 32, seed=6)
    q_lens = torch.tensor([32, 32, 20, 32, 1, 32, 32, 7, 32], dtype=torch.int32)
    scorer = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=16)
    ncells, thr, ndocs = policy
    outs = {}
    # None = no switch set: at K = 32768 that is the approximate-then-refine form of the sliced kernel, whose stage-2 list is the
    # same SET in another order (the measured-loser forms regs / ldsb live in the FLMR_EXPERIMENTAL_VARIANTS build only)
    impls = ("lds", "xcd") if K < 32768 else ("lds", "xcd", None)
    for impl in impls:
        with nat.options(**({"FLMR_S2_IMPL": impl} if impl else {})):
            for tag, ql in (("full", None), ("ragged", q_lens)):
                p, s, c = scorer.search_batch(Q, ndocs // 4, ncells, thr, ndocs, 32, q_lens=ql)
                torch.cuda.synchronize()
                taps = [scorer.tap(nat.TAP_STAGE2, q) for q in range(Q.size(0))]
                outs[impl, tag] = (p.cpu().numpy(), s.cpu().numpy(), c.cpu().numpy(), taps)
            scorer.check()
    for other in impls[1:]:
        for tag in ("full", "ragged"):
            a, b = outs[impls[0], tag], outs[other, tag]
            for q in range(Q.size(0)):
                if other is None:
                    assert sorted(a[3][q].tolist()) == sorted(b[3][q].tolist()), ("stage-2 finalists (set)", other, tag, q)
                else:
                    assert np.array_equal(a[3][q], b[3][q]), ("stage-2 finalists", other, tag, q)
            assert np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), (other, tag)


@pytest.mark.gpu
@pytest.mark.parametrize("streams", [1, 2, 3])
def test_sub_batches_over_streams_equal_one_call(hip, streams):
    """A batch cut into sub-batches (max_batch) dealt to `streams` searchers / HIP streams returns exactly what one native
    call over the whole batch returns (ids, score bits, counts), with and without ragged q_lens, from a side stream of
    the caller's too; the per-stage timing read once covers every sub-batch; taps read the last sub-batch."""
    nat = hip["native"]
    torch = hip["torch"]
    from ravqa_amd import synth
    from ravqa_amd.scorer import IndexScorer
    corpus = synth.make_corpus(30_000, (20, 150), 4096, 2, seed=47, device="cuda")
    n = 70                                          # 5 sub-batches of 16 (the last one of 6)
    Q, _ = synth.make_queries(corpus, n, 32, seed=8)
    q_lens = torch.randint(1, 33, (n,), dtype=torch.int32, generator=torch.Generator().manual_seed(3))
    di = synth.corpus_device_index(corpus)
    whole = IndexScorer(device_index=di, max_batch=128, streams=1)
    cut = IndexScorer(device_index=di, max_batch=16, streams=streams)
    ncells, thr, ndocs = 2, 0.45, 256
    for ql in (None, q_lens):
        ref = whole.search_batch(Q, ndocs // 4, ncells, thr, ndocs, 32, q_lens=ql)
        whole.check()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        for stream in (None, side):
            with torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext():
                got = cut.search_batch(Q, ndocs // 4, ncells, thr, ndocs, 32, q_lens=ql, profile=True)
                ms = cut.stage_ms()
            if stream is not None:
                torch.cuda.current_stream().wait_stream(stream)
            cut.check()
            for a, b in zip(ref, got):
                assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (streams, ql is not None, stream is not None)
            assert all(v >= 0 for v in ms.values()) and sum(ms.values()) > 0
        last = [cut.tap(nat.TAP_STAGE2, q) for q in range(6)]       # queries 64..69
        want = [whole.tap(nat.TAP_STAGE2, 64 + q) for q in range(6)]
        for a, b in zip(last, want):
            assert np.array_equal(a, b)
    with pytest.raises(nat.FlmrNativeError):
        cut.stage_ms()                              # read once: nothing profiled since
    assert cut.workspace_bytes() < whole.workspace_bytes()
    whole.close_searcher()
    cut.close_searcher()


def test_stage2_xcd_on_golden_fixture(hip, scorers):
    """The sliced stage 2 against the reference's own stage-2 output (golden `filtered_pids`, produced by filter_pids.cpp) -- as a
    set, and in order against the oracle's pruning run on the GPU's own score table."""
    from oracle import oracle as orc
    nat = hip["native"]
    z, scorer = scorers["idx_nb2"]
    oi = orc.OracleIndex.from_golden(z)
    for r in ("rank0", "rank3"):
        with nat.options(FLMR_S2_IMPL="xcd"):
            _search_one(hip, scorer, z, r, full_table=False)
            got = scorer.tap(nat.TAP_STAGE2)
            cand = scorer.tap(nat.TAP_CANDIDATES)
        _search_one(hip, scorer, z, r, full_table=True)
        table = scorer.tap(nat.TAP_CENTROID_SCORES)
        fin = oi.filter_pids(cand, table, table.max(axis=1) >= np.float32(z[f"{r}.thr"]), int(z[f"{r}.ndocs"]))
        assert np.array_equal(got, fin), r
        assert sorted(got.tolist()) == sorted(z[f"{r}.filtered_pids"].tolist()), r


def test_stage2_walk_on_golden_fixture(hip, scorers):
    """The walk against the reference's own stage-2 output (golden `filtered_pids`, produced by filter_pids.cpp) -- as a set, and
    in order against the oracle's pruning run on the GPU's own score table (like test_search_stages_vs_golden does for the
    default kernel)."""
    from oracle import oracle as orc
    nat = hip["native"]
    z, scorer = scorers["idx_nb2"]
    oi = orc.OracleIndex.from_golden(z)
    for r in ("rank0", "rank3"):
        with nat.options(FLMR_S2_IMPL="walk"):
            _search_one(hip, scorer, z, r, full_table=False)
            got = scorer.tap(nat.TAP_STAGE2)
            cand = scorer.tap(nat.TAP_CANDIDATES)
        _search_one(hip, scorer, z, r, full_table=True)
        table = scorer.tap(nat.TAP_CENTROID_SCORES)
        fin = oi.filter_pids(cand, table, table.max(axis=1) >= np.float32(z[f"{r}.thr"]), int(z[f"{r}.ndocs"]))
        assert np.array_equal(got, fin), r
        assert sorted(got.tolist()) == sorted(z[f"{r}.filtered_pids"].tolist()), r


def test_flmr_model_surface_score_and_forward(hip):
    """FLMRModelForRetrieval.score / forward (ravqa_amd/flmr.py; names of TPC/modeling/colbert.py:64-80,217-224) with
    injected encoders: the HIP padded scorer behind `score`, `forward` = query -> doc -> repeat_interleave(nway) -> score,
    against the CPU oracle's padded MaxSim on the same encodings; autograd use is refused, not silently dropped."""
    from oracle import oracle as orc
    from types import SimpleNamespace
    from ravqa_amd.flmr import FLMRModelForRetrieval
    torch = hip["torch"]
    g = torch.Generator().manual_seed(2)
    table = torch.randn(60, 128, generator=g)
    enc = lambda ids, am: (table.to(ids.device)[ids] * am.unsqueeze(-1))
    model = FLMRModelForRetrieval(enc, colbert_config=SimpleNamespace(nway=3, use_ib_negatives=True, similarity="cosine",
                                                                     interaction="colbert"), mask_punctuation_ids=[7])
    B, nway, Lq, Ld = 4, 3, 12, 20
    qi = torch.randint(1, 60, (B, Lq), generator=g)
    di = torch.randint(0, 60, (B * nway, Ld), generator=g)
    di[:, 0] = 5
    with torch.no_grad():
        scores = model.forward((qi, (qi != 0).long()), (di, (di != 0).long()))
        Q = model.query(qi, (qi != 0).long())
        D, mask = model.doc(di, (di != 0).long(), keep_dims="return_mask")
    assert scores.shape == (B * nway,) and scores.is_cuda
    ref = orc.colbert_score_padded(Q.repeat_interleave(nway, dim=0).cpu().numpy(), D.cpu().numpy(), mask.squeeze(-1).cpu().numpy())
    assert np.max(np.abs(scores.cpu().numpy() - ref)) <= SCORE_TOL
    Qg = Q.clone().requires_grad_(True)
    with pytest.raises(RuntimeError, match="forward-only"):
        model.score(Qg.repeat_interleave(nway, dim=0), D, mask)


def test_rccl_exchange_entry_points_single_rank_communicator(hip):
    """flmr_topk_allgather / flmr_keys_allgather / flmr_keys_allreduce_sum (the exchange steps of SURVEY 8e for a caller
    without torch) over a REAL RCCL communicator -- one rank, the most this box has -- created here through ctypes the way a
    C caller would (ncclGetUniqueId + ncclCommInitRank): checks the symbol resolution from the loaded RCCL, dtypes, the
    grouped gather and the merge that follows, against ops.merge_topk on the same lists."""
    import ctypes as C
    torch, ops, nat = hip["torch"], hip["ops"], hip["native"]
    lib = nat.load()
    rccl = None
    for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1",
                 os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")):
        try:
            rccl = C.CDLL(name, mode=C.RTLD_GLOBAL)
            break
        except OSError:
            continue
    assert rccl is not None, "no RCCL on this box"

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        g = torch.Generator().manual_seed(3)
        n, k = 37, 100
        scores = torch.randn(n, k, generator=g).sort(dim=1, descending=True).values.cuda()
        pids = torch.randint(0, 1 << 20, (n, k), generator=g, dtype=torch.int32).cuda()
        pids[5, 60:] = -1                                        # a short list
        gs, gp = torch.empty(1, n, k, device="cuda"), torch.empty(1, n, k, dtype=torch.int32, device="cuda")
        os_, op, oc = torch.empty(n, k, device="cuda"), torch.empty(n, k, dtype=torch.int32, device="cuda"), torch.empty(n, dtype=torch.int32, device="cuda")
        P = lambda t: C.c_void_p(t.data_ptr())
        nat.check(lib.flmr_topk_allgather(comm, 1, P(scores), P(pids), n, k, P(gs), P(gp), P(os_), P(op), P(oc), nat.stream_ptr()))
        rs, rp, rc = ops.merge_topk(scores.unsqueeze(0), pids.unsqueeze(0))
        torch.cuda.synchronize()
        assert torch.equal(gs[0], scores) and torch.equal(gp[0], pids)
        assert torch.equal(os_, rs) and torch.equal(op, rp) and torch.equal(oc, rc) and int(oc[5]) == 60
        keys = torch.randint(0, 1 << 62, (n * 64,), generator=g, dtype=torch.int64).cuda()
        out = torch.zeros_like(keys)
        nat.check(lib.flmr_keys_allgather(comm, 1, P(keys), keys.numel(), P(out), nat.stream_ptr()))
        red = keys.clone()
        nat.check(lib.flmr_keys_allreduce_sum(comm, P(red), red.numel(), nat.stream_ptr()))
        torch.cuda.synchronize()
        assert torch.equal(out, keys) and torch.equal(red, keys)
        # a communicator of another size is refused before anything is enqueued
        assert lib.flmr_keys_allgather(comm, 2, P(keys), keys.numel(), P(out), nat.stream_ptr()) != 0
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def _f16r(x):
    with np.errstate(over="ignore"):
        return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def _f16_ulps(a, b):
    a16, b16 = np.asarray(a, np.float32).astype(np.float16), np.asarray(b, np.float32).astype(np.float16)
    ia, ib = a16.view(np.int16).astype(np.int32), b16.view(np.int16).astype(np.int32)
    ia = np.where(ia < 0, -(ia & 0x7fff), ia)
    ib = np.where(ib < 0, -(ib & 0x7fff), ib)
    return np.abs(ia - ib)


@pytest.mark.parametrize("case", range(6))
def test_gpu_fp16_numerics_mode_vs_reference_expressions(hip, case):
    """FLMR_NUMERICS_GPU_FP16 (what a config with total_visible_gpus > 0 selects) against
    tests/golden/gpu_numerics.npz -- the reference's CUDA-branch torch expressions evaluated on CPU half tensors.  Stage by
    stage through the phase entry points on the FIXTURE's own intermediate lists (the branch keeps an arbitrary subset of the
    candidates tied at -inf, so only per-passage scores and above-the-cut sets are comparable across implementations), then
    end to end against the numpy restatement with this build's tie rule.  fp16 values may differ by an ulp where the fp32
    accumulation order differs; parity with the reference's CUDA kernels themselves is unpinned (see the fixture's meta)."""
    from oracle import oracle as orc
    import oracle_shard_scorer as oss
    torch, pkg, nat = hip["torch"], hip["pkg"], hip["native"]
    z = load_golden("gpu_numerics")
    zi = load_golden(str(z[f"case{case}.index"]))
    r = str(z[f"case{case}.record"])
    ncells, thr, ndocs, nqc = int(zi[f"{r}.ncells"]), float(zi[f"{r}.thr"]), int(zi[f"{r}.ndocs"]), int(zi[f"{r}.nq_cand"])
    scorer = hip["IndexScorer"](arrays=pkg.IndexArrays.from_golden(zi), numerics="gpu-fp16")
    Qn = zi[f"{r}.Q"]
    Q = torch.from_numpy(Qn).unsqueeze(0)
    k = max(ndocs // 4, 1)
    pre = f"case{case}."
    # ---- stage 0: the table holds the fp32-accumulated fp16 products; rounded they are the reference's half scores
    p, s, c = scorer.search_batch(Q, k, ncells, thr, ndocs, nqc, full_table=True)
    nq_c = min(nqc, Qn.shape[0])
    cs16 = _f16r(scorer.tap(nat.TAP_CENTROID_SCORES)[:, :nq_c])
    ref_cs = z[pre + "centroid_scores_f16"].astype(np.float32)
    assert _f16_ulps(cs16, ref_cs).max() <= 1 and (cs16 != ref_cs).mean() < 0.01
    K = cs16.shape[0]
    bits = scorer.tap(nat.TAP_IDX_BITS)
    idx = ((bits[np.arange(K) >> 5] >> (np.arange(K) & 31)) & 1).astype(bool)
    assert np.array_equal(idx, cs16.max(-1) >= _f16r(np.float32(thr)))          # half(score) >= half(thr) on its own table
    same_rows = (cs16.max(-1) == ref_cs.max(-1))
    assert np.array_equal(idx[same_rows], z[pre + "idx"][same_rows])
    cand = scorer.tap(nat.TAP_CANDIDATES)
    ref_cand = z[pre + "cand_pids"]
    if not np.array_equal(cand, ref_cand):    # only through a tie at a column's top-ncells cut
        cells, ref_cells = set(scorer.tap(nat.TAP_CELLS).tolist()), set(z[pre + "cells"].tolist())
        for cc in cells ^ ref_cells:
            assert any(np.sum(ref_cs[:, q] > ref_cs[cc, q]) < ncells and np.sum(ref_cs[:, q] >= ref_cs[cc, q]) > ncells for q in range(ref_cs.shape[1])), cc
    # ---- stage 1 (phase 1 exports this index's top-ndocs keys): per-passage scores, above-the-cut set
    unpack = lambda keys: (lambda u: ((u & np.uint64(0xFFFFFFFF)).astype(np.int64), oss.ord2f((u >> np.uint64(32)).astype(np.uint32)), u != 0))(keys.cpu().numpy().view(np.uint64))
    k1p, k1s, k1v = unpack(scorer.phase1(Q, k, ncells, thr, ndocs, nqc)[0])
    ref_s1 = dict(zip(ref_cand.tolist(), z[pre + "s1_scores"].tolist()))
    if np.array_equal(cand, ref_cand):
        mine = {int(a): float(b) for a, b, v in zip(k1p, k1s, k1v) if v}
        for pid, sc_ in mine.items():
            assert _f16_ulps(sc_, ref_s1[pid]).max() <= 1, (pid, sc_, ref_s1[pid])
        if ndocs < len(ref_cand):
            cut = np.sort(z[pre + "s1_scores"])[::-1][ndocs - 1]
            above = {pid for pid, v in ref_s1.items() if v > cut and _f16_ulps(v, cut).max() > 1}
            assert above <= set(mine), len(above - set(mine))
    # ---- stage 2 on the fixture's survivor list (phase 2 scores the members of an explicit list, slot-aligned)
    mk = lambda pids: torch.from_numpy(((np.uint64(0x80000000) << np.uint64(32)) | np.asarray(pids).astype(np.uint64)).view(np.int64)).unsqueeze(0)
    s2_in = z[pre + "s2_in_pids"]
    pad = lambda a, n: np.concatenate([a, np.zeros(n - len(a), dtype=a.dtype)])
    o2p, o2s, o2v = unpack(scorer.phase2(mk(s2_in))[0])
    assert o2v[: len(s2_in)].all() and np.array_equal(o2p[: len(s2_in)], s2_in)
    assert _f16_ulps(o2s[: len(s2_in)], z[pre + "s2_scores_f16"]).max() <= 1
    # ---- stage 3 on the fixture's finalists
    docs = z[pre + "doc_pids"]
    o3p, o3s, o3v = unpack(scorer.phase3(mk(docs))[0])
    assert o3v[: len(docs)].all() and np.array_equal(o3p[: len(docs)], docs)
    assert _f16_ulps(o3s[: len(docs)], z[pre + "doc_scores_f16"]).max() <= 2
    # ---- end to end vs the numpy restatement with the same tie rule: same finalists up to one-ulp effects
    oi = orc.OracleIndex.from_golden(zi)
    fp, fs, ncand = orc.GpuNumericsOracle(oi).rank(Qn, ncells, thr, ndocs, nqc)
    n = int(c[0])
    got_p, got_s = p[0, :n].cpu().numpy(), s[0, :n].cpu().numpy()
    assert n == len(fp) and np.array_equal(got_s, _f16r(got_s))                  # scores are fp16 values
    common = set(got_p.tolist()) & set(fp.tolist())
    assert len(common) >= 0.95 * n, (len(common), n)
    ref = dict(zip(fp.tolist(), fs.tolist()))
    for pid, sc_ in zip(got_p.tolist(), got_s.tolist()):
        if pid in ref:
            assert _f16_ulps(sc_, ref[pid]).max() <= 2, (pid, sc_, ref[pid])
    if all(_f16_ulps(fs[j], fs[j + 1]).max() > 4 for j in range(min(5, n - 1))):
        assert got_p[:5].tolist() == fp[:5].tolist()
    # the sparse-table path (scatter stage 1, recomputing stage 2) must give the same bits as the full-table path above
    p2, s2_, c2 = scorer.search_batch(Q, k, ncells, thr, ndocs, nqc)
    assert torch.equal(p2, p) and torch.equal(s2_, s) and torch.equal(c2, c)
    for impl in ("lds", "xcd"):
        with nat.options(FLMR_S2_IMPL=impl):
            p3, s3, c3 = scorer.search_batch(Q, k, ncells, thr, ndocs, nqc)
        assert torch.equal(p3, p) and torch.equal(s3, s) and torch.equal(c3, c), impl
    # stage 3 of this mode has two kernels with the same arithmetic: the wave-per-document pipeline (default for one query
    # tile; x / norm evaluated as x * (1 / norm)) and the plain one-workgroup-per-passage kernel (FLMR_S3_IMPL=f32, true
    # division): same finalists, scores equal up to an fp16 ulp
    with nat.options(FLMR_S3_IMPL="f32"):
        p4, s4, c4 = scorer.search_batch(Q, k, ncells, thr, ndocs, nqc)
    assert torch.equal(c4, c) and sorted(p4[0, :n].tolist()) == sorted(got_p.tolist())
    plain = dict(zip(p4[0, :n].tolist(), s4[0, :n].tolist()))
    d = np.array([_f16_ulps(plain[pid], sc_).max() for pid, sc_ in zip(got_p.tolist(), got_s.tolist())])
    assert d.max() <= 1 and (d == 0).mean() >= 0.9, (d.max(), (d == 0).mean())


def test_gpu_fp16_numerics_mode_through_the_searcher(hip, tmp_path):
    """Searcher(config=ColBERTConfig(total_visible_gpus=1), numerics="reference") -- the executor's single-GPU call
    (FLMR_executor.py:784) with the opt-in that follows the reference's branch selection -- runs the CUDA-branch arithmetic;
    without the opt-in, and with total_visible_gpus=0, the pinned CPU-branch arithmetic.  Both find the planted passages; the
    fp16 mode returns fp16-valued scores, also for a batch cut into sub-batches and for long queries."""
    torch, pkg = hip["torch"], hip["pkg"]
    z = load_golden("idx_nb2")
    root = str(tmp_path / "ckpt")
    pkg.IndexArrays.from_golden(z).save(os.path.join(root, "e", "indexes", "ix"))
    recs = ["rank0", "rank3", "rank9"]
    Q = torch.zeros(len(recs), 96, 128)
    for i, r in enumerate(recs):
        q = torch.from_numpy(z[f"{r}.Q"])
        Q[i, : q.size(0)] = q
    with pkg.Run().context(pkg.RunConfig(nranks=1, rank=0, root=root, experiment="e")):
        s_gpu = pkg.Searcher(index="ix", config=pkg.ColBERTConfig(total_visible_gpus=1), max_batch=2, numerics="reference")
        s_cpu = pkg.Searcher(index="ix", config=pkg.ColBERTConfig(total_visible_gpus=0), max_batch=2, numerics="reference")
        s_def = pkg.Searcher(index="ix", config=pkg.ColBERTConfig(total_visible_gpus=1), max_batch=2)
        assert s_gpu.ranker.numerics == "gpu-fp16" and s_cpu.ranker.numerics == "cpu" and s_def.numerics == "cpu"
        qs = pkg.Queries(data={i: f"q{i}" for i in range(len(recs))})
        r_gpu = s_gpu._search_all_Q(qs, Q, k=10, remove_zero_tensors=True).todict()
        r_cpu = s_cpu._search_all_Q(qs, Q, k=10, remove_zero_tensors=True).todict()
    for i, r in enumerate(recs):
        assert r_gpu[i][0][0] == r_cpu[i][0][0] == int(z[f"{r}.final_pids"][0])
        sc = np.array([t[2] for t in r_gpu[i]], dtype=np.float32)
        assert np.array_equal(sc, _f16r(sc)) and not np.array_equal(sc, np.array([t[2] for t in r_cpu[i]], dtype=np.float32))


@pytest.mark.parametrize("ties", [False, True])
def test_s0_hi_first_equals_full_products(hip, ties):
    """Stage 0 "hi first" (default: hi products everywhere, lo products only for tiles that can hold a surviving row, hi-only
    block maxima verified in the cell selection against a rigorous bound) vs the same kernels with both products everywhere
    (FLMR_S0_IMPL=f16): idx bits, cells, candidates and the final ranking must be IDENTICAL -- the shortcut changes where
    arithmetic is spent, never a stored value or a decision.  `ties`: a centroid table made of one 64-row block repeated with
    one-ulp perturbations, so that the best block maxima of every column lie within the bound of each other and the selection
    has to take its verification's slow path (all near-tying blocks recomputed; exact ties broken by the lower row index);
    cells are also checked against the oracle's selection on the full table."""
    from oracle import oracle as orc
    torch, nat = hip["torch"], hip["native"]
    from ravqa_amd import synth
    from ravqa_amd.scorer import IndexScorer
    K = 2048
    corpus = synth.make_corpus(5000, (4, 60), K, 2, seed=23, device="cuda")
    if ties:
        g = torch.Generator(device="cuda").manual_seed(7)
        base = corpus.centroids[:64].clone()
        cen = base.repeat(K // 64, 1)
        bump = (torch.rand(cen.shape, generator=g, device="cuda") < 0.02).float() * 2.0 ** -13   # ~ an fp16 ulp at |x| ~ 0.1
        bump[:64] = 0
        cen = (cen + bump).half().float()
        cen[64 * 5:64 * 6] = base                      # one exact copy: exact ties, the lower index must win
        corpus.centroids = cen.contiguous()
    scorer = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=32)
    Q, _ = synth.make_queries(corpus, 22, 32, seed=4)
    Q[3, 20:] = 0.0
    Q[7] *= 16.0                                                                  # FLMR's un-normalised visual tokens
    Q[8] *= torch.logspace(-3, 1.2, 32, device="cuda").unsqueeze(1)               # mixed magnitudes inside one query
    Q[9, ::2] *= 1e-4                                                             # rows near the fp16 subnormal range
    q_lens = torch.tensor([32, 32, 9, 20, 32, 1, 32, 32, 17, 32, 32] * 2, dtype=torch.int32)
    for (ncells, thr, ndocs) in [(1, 0.5, 64), (2, 0.45, 256), (4, 0.4, 1024), (8, 0.3, 256), (2, -1.0, 64), (2, 6.5, 256)]:
        res = {}
        # None: hi first with the flagged tiles' dense epilogue deferred to the end of the kernel (the default); "qs1": the same
        # shortcut with the dense epilogue inside the loop; "f16": both products everywhere
        for impl in (None, "qs1", "f16"):
            ctx = nat.options(FLMR_S0_IMPL=impl) if impl else contextlib.nullcontext()
            with ctx:
                p, s, c = scorer.search_batch(Q, max(ndocs // 4, 1), ncells, thr, ndocs, 32, q_lens=q_lens)
                scorer.check()
                taps = [[scorer.tap(t, q) for t in (nat.TAP_IDX_BITS, nat.TAP_CELLS, nat.TAP_CANDIDATES)] for q in range(Q.size(0))]
                res[impl] = (p.cpu().numpy(), s.cpu().numpy(), c.cpu().numpy(), taps)
        for other in ("qs1", "f16"):
            a, b = res[None], res[other]
            for q in range(Q.size(0)):
                for x, y, name in zip(a[3][q], b[3][q], ("idx", "cells", "candidates")):
                    assert np.array_equal(x, y), (ties, other, ncells, thr, q, name)
            assert np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), (ties, other, ncells, thr)
    # cells of the default path against the oracle's top-ncells (value desc, index asc) on the full table of the same kernel family
    for ncells in (2, 4):
        for q in (0, 4):
            scorer.search_batch(Q[q:q + 1], 16, ncells, 0.45, 64, 32)
            cells = scorer.tap(nat.TAP_CELLS)
            scorer.search_batch(Q[q:q + 1], 16, ncells, 0.45, 64, 32, full_table=True)
            cs = scorer.tap(nat.TAP_CENTROID_SCORES)[:, :32]
            assert np.array_equal(cells, orc.select_cells(cs, ncells)), (ties, ncells, q)


@pytest.mark.parametrize("nbits,doclen,K,npass,policy", [
    (2, (0, 200), 2048, 6000, (2, 0.45, 256)),
    (2, 64, 512, 70_000, (2, 0.3, 1024)),
    (4, (10, 90), 1000, 40_000, (4, 0.4, 4096)),
    (2, 128, 4096, 30_000, (2, 0.45, 1024)),
])
def test_stage2_approximate_then_refine_selects_the_same_set(hip, nbits, doclen, K, npass, policy):
    """Stage 2's default on the sliced kernel (FLMR_S2_IMPL=xcda): hi-only scores for every survivor, full rescoring only of
    the band around the cut, certified by the batch's error bound -- must select exactly the SET the full-score kernel selects
    (the order of the stage-2 list is not part of the contract: stage 3 rescores every member), and therefore return a final
    ranking that is bit-identical, also for short queries, empty passages and survivor counts below ndocs / 4."""
    torch, nat = hip["torch"], hip["native"]
    from ravqa_amd import synth
    from ravqa_amd.scorer import IndexScorer
    ncells, thr, ndocs = policy
    corpus = synth.make_corpus(npass, doclen, K, nbits, seed=91, device="cuda")
    scorer = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=16)
    Q, _ = synth.make_queries(corpus, 13, 32, seed=6)
    q_lens = torch.tensor([32, 32, 5, 32, 20, 32, 1, 32, 32, 32, 17, 32, 32], dtype=torch.int32)
    outs = {}
    for impl in ("xcd", "xcda"):
        with nat.options(FLMR_S2_IMPL=impl):
            p, s, c = scorer.search_batch(Q, max(ndocs // 4, 1), ncells, thr, ndocs, 32, q_lens=q_lens)
            scorer.check()
            outs[impl] = (p.cpu().numpy(), s.cpu().numpy(), c.cpu().numpy(), [scorer.tap(nat.TAP_STAGE2, q) for q in range(Q.size(0))])
    a, b = outs["xcd"], outs["xcda"]
    for q in range(Q.size(0)):
        assert sorted(a[3][q].tolist()) == sorted(b[3][q].tolist()), q
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(24))
def test_fuzz_search_vs_oracle(hip, seed):
    """Seeded random shapes through the DEFAULT path of a batched call (>= 16 queries: "hi first" stage 0 with its deferred
    pass, approximate-then-refine stage 2 where the sliced kernel runs, scatter stage 1) against the oracle: K any multiple of
    128, every nbits, ragged passages with empties, short queries via q_lens, ncells 1..4, thresholds around the policy values,
    more than one 32768-passage chunk in some cases.  Eight queries of each batch are ranked by the oracle."""
    from oracle import oracle as orc
    torch = hip["torch"]
    from ravqa_amd import synth
    from ravqa_amd.scorer import IndexScorer
    rng = np.random.default_rng(1000 + seed)
    K = 128 * int(rng.integers(4, 80))
    nbits = int(rng.choice([1, 2, 4, 8]))
    npass = int(rng.choice([3000, 9000, 20000, 40000, 70000]))
    lo = int(rng.integers(0, 12))
    doclen = (lo, lo + int(rng.integers(8, 90)))
    nqueries = int(rng.integers(16, 49))
    nq = int(rng.choice([32, 32, 24, 17]))
    ncells = int(rng.integers(1, 5))
    thr = float(rng.choice([0.3, 0.4, 0.45, 0.5, 0.55]))
    ndocs = int(rng.choice([64, 256, 1024]))
    corpus = synth.make_corpus(npass, doclen, K, nbits, seed=300 + seed, device="cuda")
    Q, _ = synth.make_queries(corpus, nqueries, nq, seed=400 + seed)
    q_lens = torch.from_numpy(rng.integers(1, nq + 1, size=nqueries).astype(np.int32))
    q_lens[::3] = nq
    arrays = synth.corpus_to_arrays(corpus)
    scorer = IndexScorer(arrays=arrays, max_batch=64)
    oi = orc.OracleIndex(arrays.dim, arrays.nbits, arrays.codes, arrays.residuals, arrays.doclens, arrays.ivf,
                         arrays.ivf_lengths, arrays.centroids, arrays.bucket_weights)
    p, s, c = scorer.search_batch(Q, ndocs // 4, ncells, thr, ndocs, 32, q_lens=q_lens)
    scorer.check()
    Qh = Q.cpu().numpy()
    for i in list(range(0, nqueries, max(1, nqueries // 8)))[:8]:
        ql = int(q_lens[i])
        rp, rs, ncand = oi.rank(Qh[i, :ql], ncells, thr, ndocs, 32)
        n = int(c[i])
        if ncand < ndocs:
            assert n == min(ncand, ndocs // 4), (seed, i, n, ncand)
            continue
        tie_aware_equal(rp, rs, p[i, :n].cpu().numpy(), s[i, :n].cpu().numpy(), tol=SCORE_TOL)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_fuzz_ops_vs_oracle(hip, seed):
    """The four extension ops (b3) on seeded random shapes against the oracle: filter_pids (ids in order, random centroid-score
    tables incl. exact ties, random idx masks, npids below and above ndocs), decompress_residuals (bytes as uint32), the
    ragged gather of rows of every element size, segmented_maxsim with empty segments."""
    from oracle import oracle as orc
    torch, ops = hip["torch"], hip["ops"]
    from ravqa_amd import synth
    rng = np.random.default_rng(9000 + seed)
    K = int(rng.integers(40, 600))
    nbits = int(rng.choice([1, 2, 4, 8]))
    npass = int(rng.integers(300, 3000))
    corpus = synth.make_corpus(npass, (0, int(rng.integers(5, 70))), max(64, (K // 64) * 64), nbits, seed=700 + seed, device="cuda")
    a = synth.corpus_to_arrays(corpus)
    K = a.num_centroids
    oi = orc.OracleIndex(a.dim, a.nbits, a.codes, a.residuals, a.doclens, a.ivf, a.ivf_lengths, a.centroids, a.bucket_weights)
    doclens = torch.from_numpy(np.asarray(a.doclens)).long()
    offsets = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(doclens, 0)])
    # ---- filter_pids ----
    nq = int(rng.choice([8, 17, 32]))
    cs = rng.standard_normal((K, nq)).astype(np.float32) * 0.3
    cs[rng.integers(0, K, size=K // 8)] = np.float32(0.25)                       # whole rows tying
    idx = cs.max(axis=1) >= np.float32(rng.choice([0.2, 0.45, 0.6]))
    for npids, ndocs in ((int(rng.integers(1, 60)), 256), (min(npass, int(rng.integers(300, 2500))), int(rng.choice([64, 256])))):
        pids = np.sort(rng.choice(npass, size=npids, replace=False)).astype(np.int32)
        want = oi.filter_pids(pids, cs, idx, ndocs)
        got = ops.filter_pids(torch.from_numpy(pids), torch.from_numpy(cs), torch.from_numpy(np.asarray(a.codes)), doclens, offsets,
                              torch.from_numpy(idx), ndocs)
        assert np.array_equal(got.cpu().numpy(), want), (seed, npids, ndocs)
    # ---- decompress_residuals ----
    pids = rng.choice(npass, size=int(rng.integers(1, 200)), replace=True).astype(np.int32)
    rev, lut = orc.codec_tables(nbits)
    D = ops.decompress_residuals(torch.from_numpy(pids), doclens, offsets, torch.from_numpy(np.asarray(a.bucket_weights)),
                                 torch.from_numpy(rev), torch.from_numpy(lut), torch.from_numpy(np.asarray(a.residuals)),
                                 torch.from_numpy(np.asarray(a.codes)), torch.from_numpy(np.asarray(a.centroids)), 128, nbits)
    assert np.array_equal(D.cpu().numpy().view(np.uint32), oi.decompress(pids).view(np.uint32)), seed
    # ---- segmented_lookup: rows of 1, 4, 8, 64, 256 bytes ----
    nseg_all = int(rng.integers(5, 400))
    lens_all = rng.integers(0, 40, size=nseg_all).astype(np.int64)
    offs_all = np.concatenate([[0], np.cumsum(lens_all)[:-1]]).astype(np.int64)
    sel = rng.integers(0, nseg_all, size=int(rng.integers(1, 300)))
    for dt, width in ((np.uint8, 1), (np.int32, 1), (np.int64, 1), (np.float32, 16), (np.float16, 128)):
        inp = rng.integers(0, 100, size=(int(lens_all.sum()) + 8, width)).astype(dt)
        if width == 1:
            inp = inp[:, 0]
        want = orc.segmented_lookup(inp, lens_all[sel], offs_all[sel])
        got = ops.segmented_lookup(torch.from_numpy(inp), torch.from_numpy(sel), torch.from_numpy(lens_all[sel]), torch.from_numpy(offs_all[sel]))
        got = got[0] if isinstance(got, tuple) else got
        assert got.cpu().numpy().tobytes() == np.asarray(want if not isinstance(want, tuple) else want[0]).tobytes(), (seed, dt, width)
    # ---- segmented_maxsim ----
    lens = rng.integers(0, 50, size=int(rng.integers(1, 120))).astype(np.int64)
    sc = rng.standard_normal((int(lens.sum()), nq)).astype(np.float32)
    out = ops.segmented_maxsim(torch.from_numpy(sc), torch.from_numpy(lens))
    want = orc.segmented_maxsim(sc, lens)
    assert np.max(np.abs(out.cpu().numpy() - want), initial=0.0) <= 1e-5 * max(1, nq)
    assert bool((out.cpu().numpy()[lens == 0] == 0.0).all())
