"""CPU oracle (oracle/flmr_oracle.c) pinned against the reference's golden vectors.

The vectors in tests/golden/*.npz were produced by running the reference CPU path
(tests/golden/make_golden.py).  Integer / table-lookup stages must be bit-exact; stages whose
reference arithmetic goes through BLAS / torch reductions carry an explicit fp32 tolerance.
"""
import numpy as np
import pytest

from conftest import load_golden, rank_records, tie_aware_equal
from oracle import oracle as orc

GEMM_TOL = 2e-6     # |centroid . q| <= 1, 128-term fp32 dot: BLAS vs k-ascending order
SCORE_TOL = 1e-4    # north_star tolerance on final fp32 scores


def test_codec_tables_match_reference(golden_index):
    _, z = golden_index
    rev, lut = orc.codec_tables(int(z["meta.nbits"]))
    assert np.array_equal(rev, z["codec.reversed_bit_map"])
    assert np.array_equal(lut, z["codec.decompression_lookup_table"])


def test_decompress_bit_exact(golden_index):
    _, z = golden_index
    oi = orc.OracleIndex.from_golden(z)
    D = oi.decompress(z["op_decompress.pids"])
    assert D.shape == z["op_decompress.D"].shape
    assert np.array_equal(D.view(np.uint32), z["op_decompress.D"].view(np.uint32))


def test_stage_chain_on_golden_inputs(golden_index):
    """Each stage fed with the reference's own upstream output: isolates stage errors."""
    _, z = golden_index
    oi = orc.OracleIndex.from_golden(z)
    for r in rank_records(z):
        Q = z[f"{r}.Q"]
        ncells, thr, ndocs = int(z[f"{r}.ncells"]), float(z[f"{r}.thr"]), int(z[f"{r}.ndocs"])
        nqc = min(int(z[f"{r}.nq_cand"]), Q.shape[0])
        cs_ref = z[f"{r}.centroid_scores"]
        # S0a: GEMM within fp32 reorder tolerance
        cs = oi.centroid_scores(Q[:nqc])
        assert np.max(np.abs(cs - cs_ref)) <= GEMM_TOL
        # S0b/S0c on the reference's scores: exact sets
        cells = orc.select_cells(cs_ref, ncells)
        assert np.array_equal(cells, z[f"{r}.cells"])
        assert np.array_equal(oi.candidates(cells), z[f"{r}.cand_pids"])
        idx = orc.idx_mask(cs_ref, thr)
        assert np.array_equal(idx, z[f"{r}.idx"])
        if f"{r}.undefined" in z:
            # reference UB (npids < ndocs): the build's defined behaviour = keep everything, no duplicates
            got = oi.filter_pids(z[f"{r}.cand_pids"], cs_ref, idx, ndocs)
            assert len(set(got.tolist())) == len(got) == min(len(z[f"{r}.cand_pids"]), ndocs // 4)
            continue
        # S1+S2 on the reference's scores: bit-exact pids in the reference's order
        fin = oi.filter_pids(z[f"{r}.cand_pids"], cs_ref, idx, ndocs)
        assert np.array_equal(fin, z[f"{r}.filtered_pids"])
        # S3a on the reference's finalists: bit-exact rows (head only is stored)
        nh = int(z[f"{r}.D_head_len"])
        D = oi.decompress(fin)
        assert np.array_equal(D[:nh].view(np.uint32), z[f"{r}.D_head"].view(np.uint32))
        Dn = orc.normalize_rows(D)
        assert np.max(np.abs(Dn[:nh] - z[f"{r}.Dn_head"])) <= 2e-7
        # S3c+d
        sc = orc.maxsim_packed(Dn, Q, oi.doclens[fin])
        assert np.max(np.abs(sc - z[f"{r}.doc_scores"])) <= SCORE_TOL


def test_rank_end_to_end(golden_index):
    _, z = golden_index
    oi = orc.OracleIndex.from_golden(z)
    for r in rank_records(z):
        if f"{r}.undefined" in z:
            continue
        Q = z[f"{r}.Q"]
        pids, scores, ncand = oi.rank(Q, int(z[f"{r}.ncells"]), float(z[f"{r}.thr"]), int(z[f"{r}.ndocs"]),
                                      int(z[f"{r}.nq_cand"]))
        assert ncand == len(z[f"{r}.cand_pids"])
        tie_aware_equal(z[f"{r}.final_pids"], z[f"{r}.final_scores"], pids, scores, tol=SCORE_TOL)


def test_segmented_maxsim_zero_clamp(golden_ops):
    z = golden_ops
    out = orc.segmented_maxsim(z["maxsim.scores"], z["maxsim.lengths"])
    assert out[0] == 0.0 and out[2] == 0.0          # all-negative doc and empty doc -> 0 (segmented_maxsim.cpp:58)
    assert np.max(np.abs(out - z["maxsim.output"])) <= 1e-5
    out = orc.segmented_maxsim(z["maxsim45.scores"], z["maxsim.lengths"])
    assert np.max(np.abs(out - z["maxsim45.output"])) <= 1e-5


@pytest.mark.parametrize("tag", ["u8", "i32", "i64", "f32", "f16"])
def test_segmented_lookup(golden_ops, tag):
    z = golden_ops
    pids = z["lookup.pids"]
    out = orc.segmented_lookup(z[f"lookup.{tag}.input"], z["lookup.lengths"][pids], z["lookup.offsets"][pids])
    assert out.tobytes() == z[f"lookup.{tag}.output"].tobytes()


def test_colbert_score_padded(golden_ops):
    z = golden_ops
    out = orc.colbert_score_padded(z["padded.Q"], z["padded.D"], z["padded.mask"])
    assert np.max(np.abs(out - z["padded.output"])) <= 1e-4 * 32
    assert out[3] == np.float32(-9999.0 * 32)       # empty doc: every token masked
    out = orc.colbert_score_padded(z["padded_aligned.Q"], z["padded.D"], z["padded.mask"])
    assert np.max(np.abs(out - z["padded_aligned.output"])) <= 1e-4 * 32


@pytest.mark.skipif(not orc.ref_available(), reason="oracle/_ref not built (needs /root/reference at build time)")
def test_restatement_vs_compiled_reference(golden_index):
    """oracle/_ref = the reference's own C++ compiled in place: restatement must agree bit-for-bit on
    filter_pids / decompress, and RefCpuScorer.rank must reproduce the golden rank()."""
    import torch
    _, z = golden_index
    oi = orc.OracleIndex.from_golden(z)
    ref = orc.RefCpuScorer(oi)
    for r in rank_records(z):
        if f"{r}.undefined" in z:
            continue
        cs_ref = z[f"{r}.centroid_scores"]
        fin = ref.filter_pids(torch.from_numpy(z[f"{r}.cand_pids"]), torch.from_numpy(cs_ref), ref.codes, ref.doclens,
                              ref.offsets, torch.from_numpy(z[f"{r}.idx"]), int(z[f"{r}.ndocs"]))
        assert np.array_equal(fin.numpy(), oi.filter_pids(z[f"{r}.cand_pids"], cs_ref, z[f"{r}.idx"], int(z[f"{r}.ndocs"])))
        pids, scores, _ = ref.rank(torch.from_numpy(z[f"{r}.Q"]), int(z[f"{r}.ncells"]), float(z[f"{r}.thr"]),
                                   int(z[f"{r}.ndocs"]), int(z[f"{r}.nq_cand"]))
        assert pids == z[f"{r}.final_pids"].tolist()
        assert np.allclose(scores, z[f"{r}.final_scores"], atol=1e-6)


def _f16_ulps(a, b):
    """distance in fp16 units-in-the-last-place between two arrays of fp16-representable values (inf == inf -> 0)"""
    a16, b16 = np.asarray(a, np.float32).astype(np.float16), np.asarray(b, np.float32).astype(np.float16)
    ia, ib = a16.view(np.int16).astype(np.int32), b16.view(np.int16).astype(np.int32)
    ia = np.where(ia < 0, -(ia & 0x7fff), ia)
    ib = np.where(ib < 0, -(ib & 0x7fff), ib)
    return np.abs(ia - ib)


def test_gpu_numerics_oracle_vs_reference_expressions():
    """SURVEY 8f-4 (FLMR_NUMERICS_GPU_FP16): the numpy restatement of the reference's CUDA-branch arithmetic
    (oracle.GpuNumericsOracle) against tests/golden/gpu_numerics.npz -- the reference's own torch expressions of that branch
    evaluated on CPU half tensors (make_golden_gpu_numerics.py).  Stage by stage on the fixture's OWN intermediate lists,
    because the branch itself is not deterministic across implementations: with fp16 scores most stage-1 candidates tie at
    -inf and torch.topk keeps an arbitrary subset of them.  fp16 values may differ by one ulp where the fp32 accumulation
    order differs (CPU half matmul vs numpy)."""
    from oracle import oracle as orc
    z = load_golden("gpu_numerics")
    assert len(z["meta.unpinned"]) >= 3          # the fixture states what could not be run with reference code
    for n in range(int(z["meta.n_cases"])):
        zi = load_golden(str(z[f"case{n}.index"]))
        r = str(z[f"case{n}.record"])
        oi = orc.OracleIndex.from_golden(zi)
        g = orc.GpuNumericsOracle(oi)
        Q, ncells, thr, ndocs, nqc = zi[f"{r}.Q"], int(zi[f"{r}.ncells"]), float(zi[f"{r}.thr"]), int(zi[f"{r}.ndocs"]), int(zi[f"{r}.nq_cand"])
        raw = g.centroid_scores_raw(Q, nqc)
        cs16 = orc.f16(raw)
        ref_cs = z[f"case{n}.centroid_scores_f16"].astype(np.float32)
        assert _f16_ulps(cs16, ref_cs).max() <= 1 and (cs16 != ref_cs).mean() < 0.01, n
        # cells: identical unless a column's cut falls inside a run of equal fp16 scores (torch.topk picks arbitrarily there)
        cells = g.cells(raw, ncells)
        ref_cells = z[f"case{n}.cells"]
        for c in set(cells.tolist()) ^ set(ref_cells.tolist()):
            assert any(np.sum(ref_cs[:, k] > ref_cs[c, k]) < ncells and np.sum(ref_cs[:, k] >= ref_cs[c, k]) > ncells for k in range(ref_cs.shape[1])), (n, c)
        idx = g.idx(ref_cs, thr)
        assert np.array_equal(idx, z[f"case{n}.idx"]), n
        cand = z[f"case{n}.cand_pids"]
        if set(cells.tolist()) == set(ref_cells.tolist()):
            assert np.array_equal(oi.candidates(cells), cand), n
        s1 = g.approx_scores(ref_cs, cand, idx)
        assert _f16_ulps(s1, z[f"case{n}.s1_scores"]).max() <= 1, n
        # stage-1 survivors: everything strictly above the cut must be kept; the rest of the list is any tie at the cut
        ref_s1 = z[f"case{n}.s1_scores"]
        kept = set(z[f"case{n}.s1_pids"].tolist())
        if ndocs < len(cand):
            cut = np.sort(ref_s1)[::-1][ndocs - 1]
            assert set(cand[ref_s1 > cut].tolist()) <= kept and all(ref_s1[np.searchsorted(cand, p)] >= cut for p in kept), n
            mine, _ = g.top(s1, cand, ndocs)
            assert set(cand[ref_s1 > cut].tolist()) <= set(mine.tolist()), n
        s2_in = z[f"case{n}.s2_in_pids"]
        s2 = g.approx_scores(ref_cs, s2_in)
        assert _f16_ulps(s2, z[f"case{n}.s2_scores_f16"]).max() <= 1, n
        docs = z[f"case{n}.doc_pids"]
        lens = zi["index.doclens"][docs[:4]]
        D = g.embeddings(docs[:4])
        assert D.shape[0] == int(lens.sum()) and _f16_ulps(D, z[f"case{n}.D_head_f16"]).max() <= 1, n
        sc = g.doc_scores(Q, docs)
        # a passage's score is a sum of Nq fp16 maxima: one ulp of a maximum can move the fp16 sum by an ulp or two
        assert _f16_ulps(sc, z[f"case{n}.doc_scores_f16"]).max() <= 2, n
