"""CPU-only checks: C-ABI library loads and exports every declared symbol, host-side mirror logic, index I/O,
sharding arithmetic, compress / IVF restatements against the reference's golden vectors."""
import os
import re

import numpy as np
import pytest
import torch

import ravqa_amd
from conftest import INDEX_FIXTURES, ROOT, load_golden
from ravqa_amd import ColBERTConfig, IndexArrays, Queries, Ranking, Run, RunConfig, _native, synth


def test_library_exports_every_declared_symbol():
    ravqa_amd.build_native()
    lib = _native.load(require_device=False)
    header = open(os.path.join(ROOT, "include", "flmr_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(flmr_[a-z0-9_]+)\s*\(", header, flags=re.M))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/flmr_hip.h but not exported"
    assert declared == set(_native.EXPORTED_SYMBOLS)
    assert lib.flmr_abi_version() == _native.ABI_VERSION == int(re.search(r"#define FLMR_ABI_VERSION (\d+)", header).group(1))
    # the compiler the binary came from is recorded beside it (the hand-scheduled kernels are verified with that toolchain)
    assert "clang version" in _native.toolchain() and "gfx950" in _native.toolchain()
    # the tap ids the host layer passes are the header's
    taps = {name: int(v) for name, v in re.findall(r"FLMR_(TAP_[A-Z0-9_]+) = (\d+)", header)}
    assert len(taps) == 10 and all(getattr(_native, name) == v for name, v in taps.items()), taps


def test_product_path_fails_loudly_without_device():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_native.FlmrNativeError):
        _native.load(require_device=True)
    from ravqa_amd.scorer import IndexScorer
    with pytest.raises(_native.FlmrNativeError):
        IndexScorer(arrays=IndexArrays.from_golden(load_golden("idx_nb1")))


@pytest.mark.parametrize("nbits", [1, 2, 4, 8])
def test_codec_tables(nbits):
    z = load_golden({1: "idx_nb1", 2: "idx_nb2", 4: "idx_nb4", 8: "idx_nb8"}[nbits])
    rev, lut = ravqa_amd.codec_tables(nbits)
    assert np.array_equal(rev, z["codec.reversed_bit_map"]) and np.array_equal(lut, z["codec.decompression_lookup_table"])


@pytest.mark.parametrize("name", INDEX_FIXTURES)
def test_compress_and_ivf_match_reference(name):
    z = load_golden(name)
    cen = torch.from_numpy(z["index.centroids_f16"].astype(np.float32))
    codes, res = synth.compress(torch.from_numpy(z["compress.embs"]), cen, torch.from_numpy(z["index.bucket_cutoffs"]), int(z["meta.nbits"]))
    assert np.array_equal(codes.numpy(), z["compress.codes"]) and np.array_equal(res.numpy(), z["compress.residuals"])
    ivf, lens = synth.build_ivf(torch.from_numpy(z["index.codes"]), torch.from_numpy(z["index.doclens"]), int(z["meta.K"]))
    assert np.array_equal(ivf.numpy(), z["index.ivf"]) and np.array_equal(lens.numpy(), z["index.ivf_lengths"])


def test_index_roundtrip_reference_format(tmp_path):
    z = load_golden("idx_nb4")
    a = IndexArrays.from_golden(z)
    a.save(str(tmp_path / "idx"))
    b = ravqa_amd.load_index_arrays(str(tmp_path / "idx"))
    for f in ("codes", "residuals", "doclens", "doc_offsets", "ivf", "ivf_lengths", "centroids", "bucket_weights"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert (b.dim, b.nbits) == (128, 4)
    cfg = ColBERTConfig.load_from_index(str(tmp_path / "idx"))
    assert cfg.nbits == 4 and cfg.dim == 128 and cfg.query_maxlen == 32


def test_legacy_ivf_conversion(tmp_path):
    z = load_golden("idx_nb1")
    a = IndexArrays.from_golden(z)
    d = str(tmp_path / "idx")
    a.save(d)
    os.remove(os.path.join(d, "ivf.pid.pt"))
    s = torch.from_numpy(a.codes).long().sort()
    torch.save((s.indices, torch.bincount(s.values, minlength=a.num_centroids)), os.path.join(d, "ivf.pt"))
    b = ravqa_amd.load_index_arrays(d)
    assert np.array_equal(a.ivf, b.ivf) and np.array_equal(a.ivf_lengths, b.ivf_lengths)


def test_loader_on_reference_written_index():
    """tests/golden/refindex_2chunk/ was written by the reference's own IndexSaver / CollectionIndexer finalize steps
    (make_refindex.py); refindex.npz is what the reference's own loader held in memory after reading it back
    (index_loader.py:31-57, residual.py:134-150, residual_embeddings.py:27-52).  Two chunks, an empty passage, ragged."""
    from conftest import GOLDEN
    from oracle import oracle as orc
    exp = dict(np.load(os.path.join(GOLDEN, "refindex.npz")))
    a = ravqa_amd.load_index_arrays(os.path.join(GOLDEN, "refindex_2chunk"))
    assert (a.dim, a.nbits, a.num_centroids) == (128, 2, 32) and int(exp["num_chunks"]) == 2
    assert a.num_embeddings == int(exp["num_embeddings"]) and a.num_passages == 90
    for f in ("codes", "residuals", "doclens", "ivf", "ivf_lengths", "centroids", "bucket_weights", "bucket_cutoffs"):
        got, want = np.asarray(getattr(a, f)), exp[f]
        if f == "ivf":  # the reference's StridedTensor keeps a zero tail behind the packed lists (strided_tensor_core.py:33-36)
            want = want[: int(exp["ivf_lengths"].sum())]
        assert got.dtype == want.dtype or f in ("bucket_cutoffs",), (f, got.dtype, want.dtype)
        assert np.array_equal(got, want), f
    assert abs(a.avg_residual - float(exp["avg_residual"])) < 1e-7
    assert a.check_ivf_invariant()
    cfg = ColBERTConfig.load_from_index(os.path.join(GOLDEN, "refindex_2chunk"))
    assert cfg.nbits == 2 and cfg.dim == 128 and cfg.checkpoint == "synthetic-no-checkpoint" and cfg.doc_maxlen == 16
    # the loaded arrays rank like the reference ranked its own load of the same directory
    oi = orc.OracleIndex(a.dim, a.nbits, a.codes, a.residuals, a.doclens, a.ivf, a.ivf_lengths, a.centroids, a.bucket_weights)
    p, s, _ = oi.rank(exp["rank.Q"], int(exp["rank.ncells"]), float(exp["rank.thr"]), int(exp["rank.ndocs"]), 32)
    from conftest import tie_aware_equal
    tie_aware_equal(exp["rank.pids"], exp["rank.scores"], p, s)


def test_loader_on_legacy_and_gpu_built_variants():
    """Same chunks, but `ivf.pt` (embedding ids, converted on load like index_loader.py:33-36 -> optimize_ivf), an fp16
    [dim] `avg_residual.pt` and fp16 `buckets.pt` (what a codec built with total_visible_gpus > 0 saves, residual.py:32-40)."""
    from conftest import GOLDEN
    exp = dict(np.load(os.path.join(GOLDEN, "refindex.npz")))
    b = ravqa_amd.load_index_arrays(os.path.join(GOLDEN, "refindex_legacy"))
    for f in ("codes", "residuals", "doclens", "ivf", "ivf_lengths", "centroids"):
        want = exp[f][: int(exp["ivf_lengths"].sum())] if f == "ivf" else exp[f]
        assert np.array_equal(np.asarray(getattr(b, f)), want), f
    assert b.bucket_weights.dtype == np.float32
    assert np.array_equal(b.bucket_weights, exp["bucket_weights"].astype(np.float16).astype(np.float32))  # widened like residual.py:44-45
    assert abs(b.avg_residual - float(np.float16(exp["avg_residual"]))) < 1e-6


def test_shard_arithmetic():
    a = IndexArrays.from_golden(load_golden("idx_nb2"))
    shards = [a.shard(r, 4) for r in range(4)]
    assert sum(s.num_passages for s in shards) == a.num_passages
    assert sum(s.num_embeddings for s in shards) == a.num_embeddings
    assert np.array_equal(np.concatenate([s.codes for s in shards]), a.codes)
    for c in (0, 17, a.num_centroids - 1):
        glob = a.ivf[a.ivf_offsets[c]:a.ivf_offsets[c + 1]]
        parts = [s.ivf[s.ivf_offsets[c]:s.ivf_offsets[c + 1]] + s.pid_base for s in shards]
        assert np.array_equal(np.concatenate(parts), glob)
    assert shards[0].pid_base == 0 and shards[3].pid_base + shards[3].num_passages == a.num_passages


def test_config_merge_semantics():
    base = ColBERTConfig(nbits=2, query_maxlen=48)
    over = ColBERTConfig(total_visible_gpus=0, nbits=None)
    m = ColBERTConfig.from_existing(base, over)
    assert m.nbits == 2 and m.query_maxlen == 48 and m.total_visible_gpus == 0
    assert "ncells" not in m.assigned and m.ncells is None
    m.configure(ncells=2)
    assert m.ncells == 2 and m.assigned["ncells"]
    with Run().context(RunConfig(nranks=1, rank=3, root="/tmp/r", experiment="temp_index_0")):
        c = ColBERTConfig.from_existing(ColBERTConfig(total_visible_gpus=0), Run().config)
        assert c.index_root_ == "/tmp/r/temp_index_0/indexes/" and c.rank == 3
    assert Run().config.experiment == "default"
    with pytest.raises(Exception):
        m.set("no_such_key", 1)


def test_data_types(tmp_path):
    q = Queries(data={7: "a", 9: {"question": "b", "answers": ["x"]}})
    assert list(q.keys()) == [7, 9] and q[9] == "b" and q.qas()[9]["answers"] == ["x"]
    r = Ranking(data={7: [(3, 1, 2.5), (4, 2, 1.5)], 9: [(1, 1, 9.0)]})
    assert r.todict()[7][1] == (4, 2, 1.5) and r.tolist()[0] == (7, 3, 1, 2.5)
    path = r.save(str(tmp_path / "out.ranking.tsv"))
    r2 = Ranking(path=path)
    assert r2.todict() == {7: [(3, 1, 2.5), (4, 2, 1.5)], 9: [(1, 1, 9.0)]}
    assert Queries.cast(["x", "y"])[1] == "y"


def test_compact_nonzero_rows_matches_reference_semantics():
    from ravqa_amd.searcher import Searcher
    z = load_golden("idx_nb2")
    Qraw = torch.from_numpy(z["rank_rz.Q_raw"]).unsqueeze(0)
    Qc, lens = Searcher._compact_nonzero_rows(torch.cat([Qraw, Qraw.flip(1)]))
    keep = torch.abs(Qraw).sum(-1) > 0            # searcher.py:120-126
    assert int(lens[0]) == int(keep.sum()) == int(lens[1])
    assert torch.equal(Qc[0, : int(lens[0])], Qraw[keep])
    assert float(Qc[0, int(lens[0]):].abs().sum()) == 0.0


def test_keys_order_like_priority_queue():
    """(score,pid) key packing used by the kernels: same order as std::pair<float,int> compare (filter_pids.cpp:24)."""
    import struct

    def f2ord(f):
        u = struct.unpack("<I", struct.pack("<f", f + 0.0))[0]
        return (~u & 0xFFFFFFFF) if (u & 0x80000000) else (u | 0x80000000)

    pairs = [(-319968.0, 5), (-319968.0, 900), (0.0, 1), (-0.0, 2), (1.5, 0), (1.5, 7), (-2.25, 3), (30.0, 2 ** 31 - 1)]
    keys = sorted(pairs, key=lambda sp: (f2ord(sp[0]) << 32) | sp[1])
    assert keys == sorted(pairs, key=lambda sp: (sp[0], sp[1]))


def test_evaluation_glue_matches_reference_semantics():
    """Ranking -> top_ranking_passages -> Recall@K, with hand-computed expectations following
    src/executors/FLMR_executor.py:852-895 and src/metrics/metrics_processors.py:481-601."""
    from ravqa_amd.evaluation import ranking_to_batch_result, recall_pseudo_relevance, recall_with_pos_ids
    contents = ["a red Bus on the road", "two cats", "the bus stop", "blue sky", "a DOG and a cat"]
    idx2id = {i: f"p{i}" for i in range(5)}
    ranking = {"q1": [(3, 1, 9.0), (0, 2, 8.0), (1, 3, 7.0)], "q2": [(1, 1, 5.0)]}   # q2 is short -> padded to max_K
    extra = {"q1": {"answers": ["bus", "train"], "gold_answer": "bus", "pos_item_ids": ["p0"]},
             "q2": {"answers": ["dog"], "gold_answer": "dog", "pos_item_ids": ["p4"]}}
    res = ranking_to_batch_result(ranking, ["q1", "q2"], idx2id, contents, max_K=3, extra_fields=extra)
    assert [p["passage_index"] for p in res[1]["top_ranking_passages"]] == [1, 1, 1]
    assert res[0]["top_ranking_passages"][1] == {"passage_index": 0, "passage_id": "p0", "content": contents[0], "score": 8.0}
    m = recall_pseudo_relevance(res, [1, 3])
    assert m["recall_at_1"] == 0.0 and m["recall_at_3"] == 0.5            # q1 finds "bus" at rank 2; q2 never finds "dog"
    assert m["precision_at_3"] == pytest.approx((1 / 3 + 0) / 2) and m["gold_recall_at_3"] == 0.5
    g = recall_with_pos_ids(res, [1, 3])
    assert g["pos_item_ids_recall_at_1"] == 0.0 and g["pos_item_ids_recall_at_3"] == 0.5
    assert g["pos_item_ids_precision_at_3"] == pytest.approx((1 / 3) / 2)


def test_build_index_end_to_end_recall_with_oracle(tmp_path):
    """Index build (k-means + compress + IVF) -> reference on-disk format -> reload -> search with the CPU oracle:
    planted queries must retrieve their target passage (the k-means cannot be bit-compared with FAISS; Recall pins it)."""
    from ravqa_amd import indexing
    from oracle import oracle as orc
    g = torch.Generator().manual_seed(0)
    K_true, P = 64, 400
    protos = torch.nn.functional.normalize(torch.randn(K_true, 128, generator=g), dim=-1)
    doclens = torch.randint(4, 24, (P,), generator=g)
    topics = torch.randint(0, K_true, (int(doclens.sum()),), generator=g)
    embs = torch.nn.functional.normalize(protos[topics] + 0.05 * torch.randn(len(topics), 128, generator=g), dim=-1)
    from host_build_backend import TorchBackend   # (host logic under test; the device steps are restated in torch)
    arrays = indexing.build_index(embs, doclens, nbits=2, num_partitions=64, kmeans_niters=6, sample_size=len(topics),
                                  backend=TorchBackend)
    arrays.save(str(tmp_path / "idx"))
    re = ravqa_amd.load_index_arrays(str(tmp_path / "idx"))
    assert re.num_centroids == 64 and re.num_embeddings == len(topics) and re.config["kmeans_niters"] == 6
    oi = orc.OracleIndex(re.dim, re.nbits, re.codes, re.residuals, re.doclens, re.ivf, re.ivf_lengths, re.centroids, re.bucket_weights)
    offs = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(doclens, 0)])
    hits = 0
    for t in range(0, 100, 5):
        toks = embs[offs[t]:offs[t + 1]]
        Q = torch.nn.functional.normalize(toks[torch.arange(32) % len(toks)] + 0.02 * torch.randn(32, 128, generator=g), dim=-1)
        pids, _, _ = oi.rank(Q.numpy(), 2, 0.45, 64)
        hits += int(t in pids[:5].tolist())
    assert hits >= 18, hits


@pytest.mark.parametrize("name", INDEX_FIXTURES)
def test_reference_indexes_satisfy_the_ivf_invariant(name):
    """The fixtures' IVF (written by the reference's optimize_ivf) lists exactly the passages that contain each centroid:
    the invariant the scatter formulation of stage 1 relies on; shards keep it; a corrupted list is detected."""
    z = load_golden(name)
    a = IndexArrays.from_golden(z)
    assert a.check_ivf_invariant()
    assert all(a.shard(r, 3).check_ivf_invariant() for r in range(3))
    bad = IndexArrays.from_golden(z)
    bad.ivf = bad.ivf.copy()
    bad.ivf[0] = (bad.ivf[0] + 1) % int(bad.doclens.shape[0])
    assert not bad.check_ivf_invariant()


def test_config_accepts_the_executors_constructor_calls():
    """FLMR_executor.py:129-134 builds ColBERTConfig(bsize=None, use_ib_negatives=True, checkpoint=..., rank=...); :786
    ColBERTConfig(total_visible_gpus=...).  The host-side mirror must take both (round 1 raised TypeError on the first)."""
    c = ColBERTConfig(bsize=None, use_ib_negatives=True, checkpoint="bert-base-uncased", rank=0)
    assert c.use_ib_negatives is True and c.bsize == 32 and "bsize" not in c.assigned and c.nway == 2
    assert ColBERTConfig(total_visible_gpus=0).total_visible_gpus == 0
    with pytest.raises(TypeError):
        ColBERTConfig(not_a_field=1)


def test_indexer_entry_point_builds_a_reference_format_index(tmp_path):
    """Indexer(checkpoint, config).index(name, collection, overwrite) (TPC/indexer.py:58-76) over indexing.build_index, on
    host tensors: path resolution under Run().context, overwrite semantics, and the written directory loads back and ranks
    the planted passages first (CPU oracle)."""
    from oracle import oracle as orc
    from ravqa_amd.indexer import Indexer
    g = torch.Generator().manual_seed(0)
    P, K = 300, 64
    protos = torch.nn.functional.normalize(torch.randn(K, 128, generator=g), dim=-1)
    doclens = torch.randint(4, 20, (P,), generator=g)
    codes = torch.randint(0, K, (int(doclens.sum()),), generator=g)
    embs = torch.nn.functional.normalize(protos[codes] + 0.05 * torch.randn(len(codes), 128, generator=g), dim=-1)
    passages = [f"passage {i}" for i in range(P)]
    calls = []

    def doc_encoder(ps):
        calls.append(len(ps))
        return embs, doclens

    with Run().context(RunConfig(nranks=1, rank=0, root=str(tmp_path), experiment="exp")):
        from host_build_backend import TorchBackend
        ix = Indexer(checkpoint=None, config=ColBERTConfig(nbits=4, kmeans_niters=4), doc_encoder=doc_encoder, build_backend=TorchBackend)
        path = ix.index("my.index", passages, overwrite=False)
        assert path == os.path.join(str(tmp_path), "exp", "indexes/", "my.index") == ix.get_index() and calls == [P]
        with pytest.raises(AssertionError):
            ix.index("my.index", passages, overwrite=False)               # exists: the reference asserts (indexer.py:67)
        assert ix.index("my.index", passages, overwrite="reuse") == path and calls == [P]      # reused, not rebuilt
        ix.index("my.index", (embs, doclens), overwrite=True)            # erased + rebuilt from an (embeddings, doclens) pair
        assert calls == [P]
        with pytest.raises(NotImplementedError):
            Indexer(checkpoint=None, config=ColBERTConfig(nbits=4), build_backend=TorchBackend).index("other", passages)
    a = ravqa_amd.load_index_arrays(path)
    assert (a.nbits, a.num_passages, a.num_embeddings) == (4, P, int(doclens.sum())) and a.check_ivf_invariant()
    oi = orc.OracleIndex(a.dim, a.nbits, a.codes, a.residuals, a.doclens, a.ivf, a.ivf_lengths, a.centroids, a.bucket_weights)
    offs = np.concatenate([[0], np.cumsum(doclens.numpy())])
    hits = 0
    for t in (3, 77, 150, 299):
        q = embs[offs[t]:offs[t + 1]][torch.arange(32) % int(doclens[t])].numpy()
        p, _, _ = oi.rank(q, 2, 0.45, 64)
        hits += int(t in p[:3].tolist())
    assert hits >= 3


def test_flmr_model_surface_host_logic():
    """FLMRModelForRetrieval (ravqa_amd/flmr.py): query / doc / mask with injected encoders vs the torch expressions of
    FLMR.query (src/models/retriever/FLMR.py:73-99) and ColBERT.doc (TPC/modeling/colbert.py:194-215) -- the parts that are
    host logic (masking, visual-token concatenation, normalisation, keep_dims); the MaxSim `score` is covered on the GPU."""
    from ravqa_amd.flmr import FLMRModelForRetrieval
    g = torch.Generator().manual_seed(1)
    B, L, dim, nvis = 3, 7, 128, 2
    table = torch.randn(50, dim, generator=g)
    enc = lambda ids, am: table[ids] * am.unsqueeze(-1)
    W = torch.randn(16, nvis * dim, generator=g)
    model = FLMRModelForRetrieval(enc, vision_projection=lambda f: f @ W, mask_punctuation_ids=[5, 9], device="cpu")
    ids = torch.tensor([[2, 5, 7, 0, 0, 0, 0], [3, 9, 9, 4, 11, 0, 0], [1, 2, 3, 4, 5, 6, 7]])
    am = (ids != 0).long()
    feats = torch.randn(B, 16, generator=g)
    Q = model.query(ids, am, feats)
    ref = torch.cat([table[ids] * am.unsqueeze(-1) * (ids != 0).unsqueeze(-1), (feats @ W).reshape(B, nvis, dim)], dim=1)
    assert Q.shape == (B, L + nvis, dim) and torch.allclose(Q, torch.nn.functional.normalize(ref, p=2, dim=2))
    assert torch.all(Q[0, 3:L].abs().sum(-1) == 0)                     # padded text tokens stay zero rows
    D, mask = model.doc(ids, am, keep_dims="return_mask")
    keep = (ids != 0) & (ids != 5) & (ids != 9)                         # punctuation ids are masked out of documents
    assert torch.equal(mask.squeeze(-1), keep) and torch.all(D[~keep].abs().sum(-1) == 0)
    assert torch.allclose(D[keep].norm(dim=-1), torch.ones(int(keep.sum())), atol=1e-6)
    flat = model.doc(ids, am, keep_dims=False)
    assert [d.shape[0] for d in flat] == keep.sum(1).tolist() and torch.allclose(flat[1], D[1][keep[1]])
    assert model.doc(ids, am).shape == (B, L, dim)
    assert model.mask(ids, skiplist=[]) == (ids != 0).tolist()


def test_metrics_match_the_reference_processors():
    """tests/golden/metrics.json: records + the numbers the REFERENCE's compute_DPR_scores / compute_DPR_scores_with_pos_ids
    (src/metrics/metrics_processors.py:481-601) computed for them (make_metrics_golden.py).  evaluation.py must reproduce
    every metric exactly, including the early-out when records carry no answers."""
    import json
    from conftest import GOLDEN
    from ravqa_amd import evaluation
    g = json.load(open(os.path.join(GOLDEN, "metrics.json")))
    got = evaluation.recall_pseudo_relevance(g["records"], g["Ks"])
    assert set(got) == set(g["pseudo_relevance"]) and len(got) == 4 * len(g["Ks"])
    for key, want in g["pseudo_relevance"].items():
        assert abs(got[key] - want) < 1e-12, key
    got = evaluation.recall_with_pos_ids(g["records"], g["Ks"])
    assert set(got) == set(g["pos_ids"])
    for key, want in g["pos_ids"].items():
        assert abs(got[key] - want) < 1e-12, key
    stripped = [{k: v for k, v in r.items() if k not in ("answers", "gold_answer")} for r in g["records"]]
    assert evaluation.recall_pseudo_relevance(stripped, g["Ks"]) == g["pseudo_relevance_without_answers"] == {}


@pytest.mark.parametrize("doclen", [16, (3, 40)])
def test_synth_shard_local_generation_equals_sliced_corpus(doclen):
    """bench.py --gpus N: every rank generates only ITS passage shard (make_corpus(pid_range=...)); the shards must be the
    slices of the one unsharded corpus (same codes, residual bytes, restricted + rebased IVF), and the planted queries --
    drawn over the whole corpus -- identical on every rank."""
    import torch
    from ravqa_amd import synth
    full = synth.make_corpus(1000, doclen, 64, 2, seed=3, chunk_tokens=4096)
    Qf, tf = synth.make_queries(full, 5, 32)
    for world in (1, 3, 8):
        for r in range(world):
            lo, hi = synth.shard_range(1000, r, world)
            a = synth.make_corpus(1000, doclen, 64, 2, seed=3, chunk_tokens=4096, pid_range=(lo, hi))
            b = synth.shard_corpus(full, r, world)
            for n in ("codes", "residuals", "doclens", "doc_offsets", "ivf", "ivf_lengths", "ivf_offsets", "bucket_weights", "centroids"):
                assert torch.equal(getattr(a, n), getattr(b, n)), (doclen, world, r, n)
            assert a.pid_base == b.pid_base == lo
            Qa, ta = synth.make_queries(a, 5, 32)
            assert torch.equal(Qa, Qf) and torch.equal(ta, tf)


def test_ranking_lists_layout():
    """Searcher.ranking_lists: device rows [n, k] -> [[(pid, rank, score)] * count] (searcher.py:81-89, :132) -- the same
    Python values as zip over Tensor.tolist(), counts honoured, ranks 1..k."""
    from ravqa_amd.searcher import Searcher
    g = torch.Generator().manual_seed(3)
    n, k = 37, 12
    P = torch.randint(0, 10_000_000, (n, k), dtype=torch.int32, generator=g)
    S = torch.rand((n, k), generator=g) * 30 - 5
    C = torch.randint(0, k + 1, (n,), dtype=torch.int32, generator=g)
    C[0], C[1] = k, 0
    want = [list(zip(P[i, :m].tolist(), range(1, k + 1), S[i, :m].tolist())) for i, m in enumerate(C.tolist())]
    for lazy in (True, False):
        got = Searcher.ranking_lists(P, S, C, k, lazy=lazy)
        assert got == want and want == [list(r) for r in got]
        assert all(type(t) is tuple and type(t[0]) is int and type(t[1]) is int and type(t[2]) is float for row in got for t in row)
        assert Searcher.ranking_lists(P[:0], S[:0], C[:0], k, lazy=lazy) == []
    # the lazy rows read like the lists they stand for, before and after the tuples exist (the executor's access patterns:
    # FLMR_executor.py:852-866 iterates and takes len(); rag_model_blip.py:402-410 indexes)
    from ravqa_amd.data import RankedList, Ranking, lazy_flat_ranking
    for fresh in (True, False):
        rows = Searcher.ranking_lists(P, S, C, k)
        assert all(isinstance(r, RankedList) for r in rows)
        if not fresh:
            for r in rows:
                list(r)
        for r, w in zip(rows, want):
            assert len(r) == len(w) and r[:5] == w[:5] and r[2:7] == w[2:7] and r[::-1] == w[::-1] and r[:0] == []
            if w:
                assert r[0] == w[0] and r[-1] == w[-1] and type(r[0]) is tuple and type(r[0][2]) is float
                pid, rank, score = r[len(w) // 2]
                assert (pid, rank, score) == w[len(w) // 2]
            with pytest.raises(IndexError):
                r[len(w)]
            assert r + [r[-1]] * 2 == w + [w[-1]] * 2 if w else r + [] == []
            assert repr(r) == repr(w) and r.tolist() == w and (r == w) and not (r != w)
            assert r.pids.tolist() == [t[0] for t in w] and r.scores.tolist() == [t[2] for t in w]
    # Ranking over lazy rows: todict() hands them through untouched, flat_ranking / save come out as before
    rows = Searcher.ranking_lists(P, S, C, k)
    qids = [f"q{i}" for i in range(n)]
    for cls in (Ranking, lazy_flat_ranking(Ranking)):
        r = cls(data=dict(zip(qids, rows)))
        d = r.todict()
        assert list(d) == qids and d["q0"] is rows[0] and d["q0"] == want[0]
        assert r.tolist() == [(q, *t) for q, w in zip(qids, want) for t in w]
        assert isinstance(r, Ranking) and cls.__name__ == "Ranking"


def test_score_reduce_interactions_and_l2_on_the_host():
    """`scoring.colbert_score_reduce` over an already computed [B, Ld, Nq] score tensor is plain torch (no device needed): the 'colbert'
    sum and the 'flipr' top-k sums of TPC/modeling/colbert.py:235-263 -- against the reference's own function when the checkout is here
    (build container), else against its restated expression -- and FLMRModelForRetrieval.score with similarity == 'l2' (colbert.py:220-222)."""
    import sys
    from types import SimpleNamespace
    from ravqa_amd import scoring
    from ravqa_amd.flmr import FLMRModelForRetrieval
    g = torch.Generator().manual_seed(3)
    B, Ld, Nq = 11, 19, 80
    scores = torch.randn(B, Ld, Nq, generator=g)
    mask = torch.rand(B, Ld, generator=g) < 0.8
    mask[:, 0] = True

    def restated(sc, m, cfg):
        sc = sc.clone()
        sc[~m] = -9999
        cm = sc.max(1).values
        if cfg.interaction == "flipr":
            out = cm[:, :64].topk(32, dim=-1).values.sum(-1)
            if 8 <= cm.size(1) - 64:
                out = out + cm[:, 64:].topk(8, dim=-1).values.sum(1)
            return out
        return cm.sum(-1)

    ref_fn = None
    ref_root = os.path.join(os.environ.get("FLMR_REFERENCE_ROOT", "/root/reference"), "third_party", "ColBERT")
    if os.path.isdir(ref_root):
        before = set(sys.modules)
        try:   # the reference's own function (imports its package: build container only; any failure leaves the restated expression)
            sys.path.insert(0, os.path.join(ROOT, "tests", "golden", "_shims"))
            sys.path.insert(0, ref_root)
            import transformers
            if not hasattr(transformers, "AdamW"):
                transformers.AdamW = torch.optim.AdamW
            from colbert.modeling.colbert import colbert_score_reduce as ref_fn   # noqa: F811
        except Exception:  # noqa: BLE001
            ref_fn = None
        finally:
            sys.path[:] = [p for p in sys.path if p not in (ref_root, os.path.join(ROOT, "tests", "golden", "_shims"))]
            # (the reference's `colbert` package and whatever the shims stood in for must not stay importable for the tests
            # after this one: test_dropin installs over a stand-in package of the same name)
            for name in set(sys.modules) - before:
                if name.split(".")[0] in ("colbert", "faiss", "ujson", "git", "bitarray", "ninja"):
                    del sys.modules[name]
    for nq in (80, 64, 70):
        for cfg in (SimpleNamespace(interaction="flipr", query_maxlen=64), SimpleNamespace(interaction="colbert", query_maxlen=64)):
            got = scoring.colbert_score_reduce(scores[:, :, :nq].clone(), mask.unsqueeze(-1), cfg)
            assert torch.equal(got, restated(scores[:, :, :nq], mask, cfg)), (nq, cfg.interaction)
            if ref_fn is not None:
                assert torch.equal(got, ref_fn(scores[:, :, :nq].clone(), mask.unsqueeze(-1), cfg)), (nq, cfg.interaction)
    # l2 similarity: the reference's torch expression, no kernel
    model = FLMRModelForRetrieval(text_encoder=None, colbert_config=SimpleNamespace(similarity="l2", interaction="colbert"), device="cpu")
    Q = torch.randn(B, 7, 16, generator=g)
    D = torch.randn(B, Ld, 16, generator=g)
    want = (-1.0 * ((Q.unsqueeze(2) - D.unsqueeze(1)) ** 2).sum(-1)).max(-1).values.sum(-1)
    assert torch.equal(model.score(Q, D, mask), want)
