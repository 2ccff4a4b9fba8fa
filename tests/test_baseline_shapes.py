"""`-m gpu` parity at BASELINE.json's shapes, against the reference's own compiled C++ stages where they are available
(oracle/_ref via RefCpuScorer -- the .so files travel to the GPU box) and the C restatement otherwise.

  cfg2  10 k passages x 128 tokens, K = 16384                    (BASELINE.json configs[1])
  cfg3  160 k passages, K = 65536, nbits = 8, Nq in {320, 832}  (configs[2]: PreFLMR's long queries + PQ decompress)
  cfg4  1 M passages x 128 tokens, K = 131072, nbits 2 and 8, fixed and ragged doclens   (configs[3], one GPU's view);
        the same corpus cut into 8 passage shards through the exact three-exchange protocol (bit-identical to unsharded)
  cfg5  6 M passages x 128 tokens (768 M tokens), K = 262144 (collection_indexer.py:93), nbits = 2 on one device, and one
        GPU's shard of it (750 k passages, K = 262144, nbits = 8)                         (configs[4])

Policies follow searcher.py:92-118: k <= 100 -> (ncells 2, thr 0.45, ndocs 1024); k = 500 -> (4, 0.4, 4096).
Bars: ranked ids identical position by position except inside runs of reference scores closer than `gap` (another valid
fp32 summation order may swap those); scores within `tol`.  For Nq = 32: gap 1e-5, tol 1e-4 (north_star).  For
long queries the score itself grows with Nq (hundreds), and one fp32 ulp of the score exceeds 1e-5, so both bars are
stated in ulps of the score magnitude: gap = 4 ulp, tol = 16 ulp (>= the Nq = 32 values).
"""
import numpy as np
import pytest

from conftest import tie_aware_equal

pytestmark = pytest.mark.gpu

POLICY = {100: (2, 0.45, 1024), 500: (4, 0.4, 4096)}   # searcher.py:92-118
EPS32 = float(np.finfo(np.float32).eps)


@pytest.fixture(scope="module")
def hip():
    import torch
    import ravqa_amd
    from ravqa_amd import _native
    assert torch.cuda.is_available(), "these tests need the MI355X"
    _native.load(require_device=True)
    return dict(torch=torch, pkg=ravqa_amd)


def _reference_ranker(arrays):
    """rank(Q numpy [Nq,128], ncells, thr, ndocs) -> (pids, scores, ncand): the reference's compiled stages when
    oracle/_ref is present, else the C restatement (pinned to them by tests/test_oracle_golden.py)."""
    import torch
    from oracle import oracle as orc
    oi = orc.OracleIndex(arrays.dim, arrays.nbits, arrays.codes, arrays.residuals, arrays.doclens, arrays.ivf,
                         arrays.ivf_lengths, arrays.centroids, arrays.bucket_weights)
    if orc.ref_available():
        ref = orc.RefCpuScorer(oi)
        return lambda Q, ncells, thr, ndocs: ref.rank(torch.from_numpy(Q), ncells, thr, ndocs), "reference"
    return lambda Q, ncells, thr, ndocs: oi.rank(Q, ncells, thr, ndocs, 32), "port"


def _check(hip, corpus, nq, n_queries, ks, seed=2, max_batch=32, local_queries=False):
    """local_queries: `corpus` is one passage shard (pid_base > 0) searched on its own -- plant the queries inside it and
    compare LOCAL pids with the reference run on the shard's arrays (flmr_search_batch returns global pids)."""
    torch = hip["torch"]
    from ravqa_amd import synth
    from ravqa_amd.scorer import IndexScorer
    scorer = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=max_batch)
    info = scorer.device_index.info()
    print("index info:", info)
    # the sliced stage 2 is the default at these table sizes when the dispatch probe confirms the XCD mapping; a slice must
    # fit an L2 (4 MB + slack) whatever K is
    if corpus.K * 256 > (6 << 20):
        assert info["stage2_sliced"] == info["xcd_round_robin"] and corpus.K * 256 <= info["stage2_slices"] * ((4 << 20) + (1 << 18)), info
    rank, kind = _reference_ranker(synth.corpus_to_arrays(corpus))
    base = 0
    if local_queries:
        import copy
        view = copy.copy(corpus)
        view.g_doclens, view.g_doc_offsets, view.g_codes = corpus.doclens, corpus.doc_offsets, corpus.codes
        base = corpus.pid_base
        Q, targets = synth.make_queries(view, max(n_queries.values()), nq, seed=seed)
    else:
        Q, targets = synth.make_queries(corpus, max(n_queries.values()), nq, seed=seed)
    Qh = Q.cpu().numpy()
    checked = 0
    for k in ks:
        ncells, thr, ndocs = POLICY[k]
        n = n_queries[k]
        p, s, c = scorer.search_batch(Q[:n], ndocs // 4, ncells, thr, ndocs, 32)
        scorer.check()
        p, s, c = p.cpu().numpy() - base, s.cpu().numpy(), c.cpu().numpy()
        for i in range(n):
            rp, rs, ncand = rank(Qh[i], ncells, thr, ndocs)
            if ncand < ndocs:       # the reference's undefined case (filter_pids.cpp:119-123): defined here, tested elsewhere
                continue
            mag = max(1.0, float(np.max(np.abs(rs))))
            gap, tol = max(1e-5, 4 * EPS32 * mag), max(1e-4, 16 * EPS32 * mag)
            m = int(c[i])
            assert m == len(rp) == ndocs // 4, (k, i, m, len(rp))
            tie_aware_equal(rp, rs, p[i, :m], s[i, :m], gap=gap, tol=tol)
            if all(abs(rs[j] - rs[j + 1]) > gap for j in range(5)):   # no near-tie among the first six: top-5 ids bit-exact
                assert p[i, :5].tolist() == [int(x) for x in rp[:5]], (k, i)
            checked += 1
        if k <= 100:
            hit = (torch.from_numpy(p[:, :5].astype(np.int64)) == targets[:n].cpu().unsqueeze(1)).any(dim=1).float().mean()
            assert float(hit) >= 0.95, float(hit)
    assert checked >= sum(n_queries[k] for k in ks) // 2, (checked, kind)
    scorer.close_searcher()
    return kind


@pytest.mark.parametrize("nbits", [2, 8])
def test_cfg2_10k_passages(hip, nbits):
    from ravqa_amd import synth
    corpus = synth.make_corpus(10_000, 128, 16384, nbits, seed=11, device="cuda")
    _check(hip, corpus, 32, {100: 32, 500: 8}, ks=(100, 500))


@pytest.mark.parametrize("nq", [320, 832])
def test_cfg3_160k_passages_long_queries_nbits8(hip, nq):
    """PreFLMR-sized queries (Nq = 320 / 832 > query_maxlen = 32): candidate generation uses the first 32 tokens
    (index_storage.py:77), the exact MaxSim all of them -- the LDS-chunked long-query kernel at the size it exists for."""
    from ravqa_amd import synth
    corpus = synth.make_corpus(160_000, 128, 65536, 8, seed=12, device="cuda")
    _check(hip, corpus, nq, {100: 8}, ks=(100,), max_batch=8)


@pytest.mark.parametrize("nbits,doclen", [(2, 128), (8, 128), (2, (32, 224)), (8, (32, 224))])
def test_cfg4_1m_passages_headline_shape(hip, nbits, doclen):
    """BASELINE's headline corpus on one GPU: 1 M passages, K = 131072 -- 32 queries at the k <= 100 policy and 4 at the
    k = 500 policy, every ranked list checked against the reference's CPU stages."""
    from ravqa_amd import synth
    corpus = synth.make_corpus(1_000_000, doclen, 131072, nbits, seed=0, device="cuda")
    _check(hip, corpus, 32, {100: 32, 500: 4}, ks=(100, 500))


def test_cfg5_6m_passages_768m_tokens_one_device(hip):
    """BASELINE configs[4] at its size, on ONE device (the 8-GPU job holds an eighth of it per rank): 6 M passages x 128
    tokens = 768 M tokens, K = 2^18 (collection_indexer.py:93), nbits = 2 -- 3 GB of codes, 24.6 GB of residual bytes.
    Exercises what the smaller shapes cannot: token positions past 2^29 (byte offsets past 2^31 in every code / residual
    access), 184 passage chunks per query in candidate generation, a 64 MB fp16 centroid table (16 stage-2 slices so that
    each still fits an XCD's L2), 190 k candidates per query.  Every ranked list is compared with the reference's CPU stages."""
    from ravqa_amd import synth
    corpus = synth.make_corpus(6_000_000, 128, 262144, 2, seed=0, device="cuda")
    assert corpus.codes.numel() == 768_000_000
    _check(hip, corpus, 32, {100: 8, 500: 2}, ks=(100, 500), max_batch=8)


def test_cfg5_one_shard_view_nbits8(hip):
    """What ONE rank of the 8-GPU configs[4] job holds: passages [2.25 M, 3 M) of the 6 M corpus (generated shard-locally),
    K = 2^18 replicated, nbits = 8 (the FLMR configs' setting: 128 residual bytes per token, 12.3 GB)."""
    from ravqa_amd import synth
    corpus = synth.make_corpus(6_000_000, 128, 262144, 8, seed=0, device="cuda", pid_range=synth.shard_range(6_000_000, 3, 8))
    assert corpus.pid_base == 2_250_000 and corpus.doclens.numel() == 750_000
    _check(hip, corpus, 32, {100: 8, 500: 2}, ks=(100, 500), max_batch=8, local_queries=True)


def _exact_protocol_on_one_device(torch, ops, shards, Q, k, ncells, thr, ndocs, split_stage0=True, truncate=True, fallback=False):
    """All ranks' phases of ShardedSearcher.search_batch_exact run in sequence on ONE device: the all-gathers are
    torch.stack / cat, the SUM all-reduces a sum over the stack (exactly the data movement of distributed.py)."""
    W, B = len(shards), Q.size(0)
    if split_stage0:
        per = -(-B // W)
        parts = []
        for r, sh in enumerate(shards):                                     # (k = 500: 4 queries over 8 ranks -> empty slices)
            lo = min(B, r * per)
            iw, mc = sh.probe_dims(Q, k, ncells, thr, ndocs, 32)
            bufs = (torch.zeros((per, iw), dtype=torch.int32, device="cuda"), torch.zeros((per, mc), dtype=torch.int32, device="cuda"),
                    torch.zeros((per,), dtype=torch.int32, device="cuda"))  # what distributed.py gathers: `per` rows per rank
            parts.append(sh.probe(Q, k, ncells, thr, ndocs, lo, min(B, lo + per) - lo, 32, out=bufs))
        bits, cells, ncell = (torch.cat([p_[j] for p_ in parts]) for j in range(3))
        k1 = [sh.phase1_probed(Q, k, ncells, thr, ndocs, bits, cells, ncell, 32) for sh in shards]
    else:
        k1 = [sh.phase1(Q, k, ncells, thr, ndocs, 32) for sh in shards]
    from ravqa_amd.distributed import merge_truncated, phase1_width
    m = phase1_width(ndocs, W) if truncate else ndocs
    if m < ndocs:   # each shard ships its m best keys + the certificate (distributed.py: the default exchange)
        g = torch.stack([ops.topn_keys(k_, m, ordered=False) for k_ in k1])                     # [W, B, m]
        s1, violated = merge_truncated(g, ndocs, ops.topn_keys)
        if fallback and bool(violated):   # skewed shards: what search_batch_exact(check=True) does -- the full exchange
            s1 = ops.topn_keys(torch.stack(k1).permute(1, 0, 2).reshape(B, -1), ndocs, ordered=False)
        else:
            assert not bool(violated)
    else:
        g = torch.stack(k1)                                                 # [W, B, ndocs]
        s1 = ops.topn_keys(g.permute(1, 0, 2).reshape(B, -1), ndocs, ordered=False)
    parts2 = torch.stack([sh.phase2(s1) for sh in shards])
    assert int((parts2 != 0).sum(dim=0).max()) <= 1                         # one contributor per slot: SUM == gather
    s2 = ops.topn_keys(parts2.sum(dim=0), ndocs // 4, ordered=False)
    parts3 = torch.stack([sh.phase3(s2) for sh in shards])
    assert int((parts3 != 0).sum(dim=0).max()) <= 1
    fin = ops.topn_keys(parts3.sum(dim=0), min(k, ndocs // 4), ordered=True)
    return ops.unpack_keys(fin, k)


def test_cfg4_sharded_into_8_exact_protocol_equals_unsharded(hip):
    """BASELINE configs[3] at its real shape: the 1 M x 128 corpus (K = 131072) cut into EIGHT passage shards, every rank's
    probe -> phase1_probed -> phase2 -> phase3 executed on this one device with stack / sum as the exchanges; the result must
    be BIT-IDENTICAL (ids, scores, counts) to the unsharded search_batch -- 32 queries at the k = 100 policy, 4 at k = 500 --
    also with the replicated stage 0; and the fast mode (one all-gather of per-shard top-k) must return exact scores over a
    superset of the unsharded survivors.  (src/executors/FLMR_executor.py:778-783 is what this replaces: the reference drops
    to its CPU path whenever world_size > 1.)"""
    torch = hip["torch"]
    from ravqa_amd import ops, synth
    from ravqa_amd.scorer import IndexScorer
    W = 8
    corpus = synth.make_corpus(1_000_000, 128, 131072, 2, seed=0, device="cuda")
    single = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=32)
    shard_corpora = [synth.shard_corpus(corpus, r, W) for r in range(W)]
    # one shard generated locally (what bench.py --gpus 8 does on each rank) is the same shard
    own = synth.make_corpus(1_000_000, 128, 131072, 2, seed=0, device="cuda", pid_range=synth.shard_range(1_000_000, 5, W))
    for n in ("codes", "residuals", "ivf", "ivf_lengths", "doc_offsets"):
        assert torch.equal(getattr(own, n), getattr(shard_corpora[5], n)), n
    del own
    shards = [IndexScorer(device_index=synth.corpus_device_index(sc), max_batch=32) for sc in shard_corpora]
    Q, targets = synth.make_queries(corpus, 32, 32, seed=2)
    for k, n in ((100, 32), (500, 4)):
        ncells, thr, ndocs = POLICY[k]
        p_ref, s_ref, c_ref = single.search_batch(Q[:n], k, ncells, thr, ndocs, 32)
        single.check()
        assert int(c_ref.min()) == k
        for split, trunc in ((True, True), (False, True), (True, False)):
            p, s, c = _exact_protocol_on_one_device(torch, ops, shards, Q[:n], k, ncells, thr, ndocs, split_stage0=split, truncate=trunc)
            for sh in shards:
                sh.check()
            assert torch.equal(c, c_ref) and torch.equal(p, p_ref) and torch.equal(s, s_ref), (k, split, trunc)
        # fast mode: every shard prunes with the same ndocs, one gather of the per-shard top-k, merged
        loc = [sh.search_batch(Q[:n], k, ncells, thr, ndocs, 32) for sh in shards]
        ms, mp, mc = ops.merge_topk(torch.stack([l[1] for l in loc]), torch.stack([l[0] for l in loc]))
        assert int(mc.min()) == k
        assert bool((ms >= s_ref).all())                      # position by position at least the exact mode's score
        for q in range(n):
            fast = dict(zip(mp[q].tolist(), ms[q].tolist()))
            kth = float(ms[q, -1])
            for pid, sc_ in zip(p_ref[q].tolist(), s_ref[q].tolist()):
                # an unsharded survivor also survives in its shard (fewer competitors): same exact score, and it is in the
                # merged list unless k superset documents outrank it
                assert (pid in fast and fast[pid] == sc_) or sc_ <= kth, (k, q, pid)
    for sh in shards + [single]:
        sh.close_searcher()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(16))
def test_fuzz_sharded_exact_protocol_equals_unsharded(hip, seed):
    """Seeded random shapes and shard counts (2..8, passage counts that do not divide evenly, shards smaller than ndocs, ragged
    passages with empties, short queries): every rank's probe -> phase1_probed -> phase2 -> phase3 on this one device with the
    exchanges of distributed.py, query-split or replicated stage 0, truncated or full phase-1 exchange -- bit-identical to the
    unsharded search_batch."""
    torch = hip["torch"]
    from ravqa_amd import ops, synth
    from ravqa_amd.scorer import IndexScorer
    rng = np.random.default_rng(7000 + seed)
    W = int(rng.integers(2, 9))
    K = 128 * int(rng.integers(8, 48))
    nbits = int(rng.choice([2, 4, 8]))
    npass = int(rng.choice([2500, 7001, 20003, 50007]))
    lo = int(rng.integers(0, 10))
    doclen = (lo, lo + int(rng.integers(10, 80)))
    B = int(rng.integers(5, 40))
    k, ncells, thr, ndocs = [(100, 2, 0.45, 1024), (10, 1, 0.5, 64), (40, 3, 0.4, 256)][int(rng.integers(0, 3))]
    corpus = synth.make_corpus(npass, doclen, K, nbits, seed=500 + seed, device="cuda")
    single = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=64)
    shards = [IndexScorer(device_index=synth.corpus_device_index(synth.shard_corpus(corpus, r, W)), max_batch=64) for r in range(W)]
    Q, _ = synth.make_queries(corpus, B, 32, seed=600 + seed)
    p_ref, s_ref, c_ref = single.search_batch(Q, k, ncells, thr, ndocs, 32)
    single.check()
    split = bool(rng.integers(0, 2))
    trunc = bool(rng.integers(0, 2))
    p, s, c = _exact_protocol_on_one_device(torch, ops, shards, Q, k, ncells, thr, ndocs, split_stage0=split, truncate=trunc, fallback=True)
    for sh in shards:
        sh.check()
    assert torch.equal(c, c_ref) and torch.equal(p, p_ref) and torch.equal(s, s_ref), (seed, W, K, nbits, npass, B, k, split, trunc)
    for sh in shards + [single]:
        sh.close_searcher()


@pytest.mark.parametrize("topics,min_survivors", [(16, 400), (256, 40)])
def test_built_index_overlapping_clusters_vs_reference(hip, topics, min_survivors):
    """The NON-planted regime: raw embeddings whose clusters overlap, indexed end to end on the device
    (k-means with the HIP argmax, compression, IVF), then 32 planted queries whose ranked lists are compared with the reference's
    compiled CPU stages on the SAME built index -- hundreds to thousands of centroids pass the threshold per query (the code-
    scanning / dense-fallback stage-1 forms), residuals are structured, passages share centroids."""
    torch = hip["torch"]
    from ravqa_amd import _native, indexing, synth
    from ravqa_amd.scorer import IndexScorer
    embs, doclens, planted = synth.make_overlapping_embeddings(40_000, 64, topics, seed=3, device="cuda", sub_directions=8192)
    arrays = indexing.build_index(embs, doclens, nbits=2, kmeans_niters=4)
    scorer = IndexScorer(arrays=arrays, max_batch=32)
    rank, kind = _reference_ranker(arrays)
    Q, tgt = planted(32)
    ncells, thr, ndocs = POLICY[100]
    p, s, c = scorer.search_batch(Q, ndocs // 4, ncells, thr, ndocs, 32)
    scorer.check()
    surv = [sum(bin(int(x) & 0xffffffff).count("1") for x in scorer.tap(_native.TAP_IDX_BITS, q)) for q in range(0, 32, 8)]
    assert max(surv) >= min_survivors, surv
    p, s, c = p.cpu().numpy(), s.cpu().numpy(), c.cpu().numpy()
    Qh = Q.cpu().numpy()
    checked = 0
    for i in range(32):
        rp, rs, ncand = rank(Qh[i], ncells, thr, ndocs)
        if ncand < ndocs:
            continue
        m = int(c[i])
        assert m == len(rp), (i, m, len(rp))
        tie_aware_equal(rp, rs, p[i, :m], s[i, :m], gap=1e-5, tol=1e-4)
        checked += 1
    assert checked >= 16, (checked, kind, surv)
    hit = (torch.from_numpy(p[:, :5].astype(np.int64)) == tgt.cpu().unsqueeze(1)).any(dim=1).float().mean()
    assert float(hit) >= 0.9, float(hit)
    scorer.close_searcher()
