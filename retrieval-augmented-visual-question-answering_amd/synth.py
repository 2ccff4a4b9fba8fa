"""Index-side data producers: residual compression and IVF construction in the reference's wire format, plus
the synthetic clustered corpus of SURVEY 8d / BASELINE.md section 3 (1 M passages x 128 tokens x 128-d).

`compress` restates ResidualCodec.compress / binarize (TPC/indexing/codecs/residual.py:169-204): nearest centroid by
dot product, residual = emb - centroid, torch.bucketize against the bucket cut-offs, bucket index emitted LSB-first
per value and packed MSB-first (np.packbits order).  `build_ivf` restates the IVF construction
(TPC/indexing/collection_indexer.py:388-426 + TPC/indexing/utils.py:8-53): per centroid the sorted unique pids.
Both are plain torch ops that run on whatever device the inputs live on; they exist so the GPU box can build the
benchmark index by itself (the reference's indexer needs FAISS + CUDA) and are pinned by golden vectors in
tests/test_host_logic.py.  They are the seed of the "index build" row of SURVEY 8f, not a tuned kernel yet.
"""
import math

import numpy as np
import torch


def pack_buckets(buckets, nbits):
    """uint8 bucket indices [N, dim] -> packed bytes [N, dim*nbits/8] (residual.py:186-204)."""
    vpb = 8 // nbits
    b = buckets.to(torch.int32)
    rev = torch.zeros_like(b)
    for j in range(nbits):  # LSB-first emission followed by MSB-first packing = bit reversal inside each group
        rev |= ((b >> j) & 1) << (nbits - 1 - j)
    rev = rev.view(b.size(0), -1, vpb)
    out = torch.zeros(rev.shape[:2], dtype=torch.int32, device=b.device)
    for l in range(vpb):
        out |= rev[:, :, l] << (8 - nbits * (l + 1))
    return out.to(torch.uint8)


def compress(embs, centroids, bucket_cutoffs, nbits, codes=None):
    """-> (codes int32 [N], residual bytes uint8 [N, dim*nbits/8])."""
    if codes is None:
        codes = (centroids @ embs.T).max(dim=0).indices
    res = embs - centroids[codes.long()]
    buckets = torch.bucketize(res.float(), bucket_cutoffs.to(res.device)).to(torch.uint8)
    return codes.to(torch.int32), pack_buckets(buckets, nbits)


def bucket_tables(residual_sample, nbits):
    """bucket_cutoffs / bucket_weights as quantiles of held-out residuals (collection_indexer.py:303-308)."""
    n = 2 ** nbits
    q = torch.arange(0, n, device=residual_sample.device, dtype=torch.float32) / n
    flat = residual_sample.float().flatten()
    if flat.numel() > 4_000_000:  # torch.quantile input limit; a FIXED subsample, so every caller derives the same tables
        flat = flat[:: -(-flat.numel() // 4_000_000)]
    return flat.quantile(q[1:]), flat.quantile(q + 0.5 / n)


def build_ivf(codes, doclens, K):
    """-> (ivf pids int32 [sum unique], ivf_lengths int64 [K])."""
    n = doclens.numel()
    pid_of = torch.repeat_interleave(torch.arange(n, device=codes.device), doclens.to(codes.device))
    key = torch.unique(codes.long() * n + pid_of)
    return (key % n).to(torch.int32), torch.bincount(key // n, minlength=K).long()


class SyntheticCorpus:
    pass


def _chunk_generator(seed, chunk, device):
    g = torch.Generator(device=device)
    g.manual_seed((int(seed) * 1_000_003 + 7919 * (int(chunk) + 1)) & 0x7FFFFFFFFFFF)
    return g


def make_corpus(n_passages, doclen, K, nbits, seed=0, device="cpu", sigma=0.05, chunk_tokens=1 << 21, dim=128, pid_range=None):
    """Clustered corpus: protos = normalize(N(0,I)[K,dim]); token = normalize(protos[c] + sigma*N(0,I)), c ~ U{0..K-1};
    the token's code is its generating centroid.  `doclen` is an int (fixed) or (lo, hi) inclusive (ragged).
    Returns a SyntheticCorpus whose tensors live on `device`.

    The cheap global state (centroids, doclens, one code per token) comes from ONE generator and is identical for every
    caller; the expensive part -- the tokens' noise, their residual bytes and the IVF -- is drawn per `chunk_tokens`-token
    chunk from a generator seeded by (seed, chunk index), so `pid_range=(lo, hi)` builds exactly the passage shard
    [lo, hi) of the same corpus (same bytes as `shard_corpus(make_corpus(...), ...)`) at 1/N of the cost: each rank of a
    sharded job generates its own shard.  A shard keeps the GLOBAL doclens / codes as `g_*` for
    `make_queries` (queries are planted over the whole corpus) and its pid offset as `pid_base`."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    protos = torch.nn.functional.normalize(torch.randn(K, dim, generator=g, device=device), dim=-1)
    centroids = protos.half().float()  # the index stores fp16 centroids (residual.py:161)
    if isinstance(doclen, int):
        doclens = torch.full((n_passages,), doclen, dtype=torch.int64, device=device)
    else:
        doclens = torch.randint(doclen[0], doclen[1] + 1, (n_passages,), generator=g, device=device, dtype=torch.int64)
    offsets = torch.cat([torch.zeros(1, dtype=torch.int64, device=device), torch.cumsum(doclens, 0)])
    N = int(offsets[-1])
    codes = torch.randint(0, K, (N,), generator=g, device=device, dtype=torch.int32)
    lo, hi = (0, n_passages) if pid_range is None else (int(pid_range[0]), int(pid_range[1]))
    assert 0 <= lo <= hi <= n_passages, (lo, hi, n_passages)
    tlo, thi = int(offsets[lo]), int(offsets[hi])
    residuals = torch.empty((thi - tlo, dim * nbits // 8), dtype=torch.uint8, device=device)

    def chunk_embeddings(ci):
        t0, t1 = ci * chunk_tokens, min(N, (ci + 1) * chunk_tokens)
        c = codes[t0:t1].long()
        noise = torch.randn(t1 - t0, dim, generator=_chunk_generator(seed, ci, device), device=device)
        return t0, t1, c, torch.nn.functional.normalize(centroids[c] + sigma * noise, dim=-1)

    # bucket tables: quantiles of the first 2^15 residuals of chunk 0 (every shard derives the same ones)
    t0, t1, c, emb = chunk_embeddings(0)
    cut, wts = bucket_tables((emb - centroids[c])[: 1 << 15], nbits)
    for ci in range(tlo // chunk_tokens, -(-thi // chunk_tokens) if thi > tlo else 0):
        if ci > 0:
            t0, t1, c, emb = chunk_embeddings(ci)
        a, b = max(t0, tlo), min(t1, thi)
        _, res = compress(emb[a - t0:b - t0], centroids, cut, nbits, codes=c[a - t0:b - t0])
        residuals[a - tlo:b - tlo] = res
    del emb, c
    out = SyntheticCorpus()
    out.dim, out.nbits, out.K, out.sigma = dim, nbits, K, sigma
    out.centroids = centroids
    out.bucket_cutoffs, out.bucket_weights = cut, wts
    out.g_doclens, out.g_doc_offsets, out.g_codes = doclens, offsets, codes
    out.pid_base, out.n_passages_global = lo, n_passages
    if pid_range is None:
        out.doclens, out.doc_offsets, out.codes = doclens, offsets, codes
    else:
        out.doclens = doclens[lo:hi].contiguous()
        out.doc_offsets = (offsets[lo:hi + 1] - tlo).contiguous()
        out.codes = codes[tlo:thi].contiguous()
    out.residuals = residuals
    out.ivf, out.ivf_lengths = build_ivf(out.codes, out.doclens, K)
    out.ivf_offsets = torch.cat([torch.zeros(1, dtype=torch.int64, device=device), torch.cumsum(out.ivf_lengths, 0)])
    return out


def shard_corpus(corpus, rank, world):
    """Passage shard `rank` of `world` of an UNSHARDED corpus: contiguous pid range, codes / residuals /
    doclens sliced, IVF restricted to the range and rebased.  Same arrays as make_corpus(..., pid_range=shard_range(...))."""
    P, K = corpus.doclens.numel(), corpus.K
    lo, hi = shard_range(P, rank, world)
    dev = corpus.codes.device
    tlo, thi = int(corpus.doc_offsets[lo]), int(corpus.doc_offsets[hi])
    keep = (corpus.ivf >= lo) & (corpus.ivf < hi)
    owner = torch.repeat_interleave(torch.arange(K, device=dev), corpus.ivf_lengths)
    sh = SyntheticCorpus()
    sh.dim, sh.nbits, sh.K, sh.sigma = corpus.dim, corpus.nbits, K, corpus.sigma
    sh.centroids, sh.bucket_weights, sh.bucket_cutoffs = corpus.centroids, corpus.bucket_weights, corpus.bucket_cutoffs
    sh.g_doclens, sh.g_doc_offsets, sh.g_codes = corpus.doclens, corpus.doc_offsets, corpus.codes
    sh.pid_base, sh.n_passages_global = lo, P
    sh.codes, sh.residuals = corpus.codes[tlo:thi].contiguous(), corpus.residuals[tlo:thi].contiguous()
    sh.doclens = corpus.doclens[lo:hi].contiguous()
    sh.doc_offsets = (corpus.doc_offsets[lo:hi + 1] - tlo).contiguous()
    sh.ivf = (corpus.ivf[keep] - lo).to(torch.int32).contiguous()
    sh.ivf_lengths = torch.bincount(owner[keep], minlength=K).long()
    sh.ivf_offsets = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(sh.ivf_lengths, 0)])
    return sh


def shard_range(n_passages, rank, world):
    return (n_passages * rank) // world, (n_passages * (rank + 1)) // world


def make_queries(corpus, n_queries, nq, seed=2, sigma=None):
    """Planted queries: query i targets passage t_i (a GLOBAL pid, also when `corpus` is a shard); its token j sits near the
    centroid of token (j mod doclen) of t_i.  Returns (Q [n, nq, dim] on the corpus device, target pids [n])."""
    doclens, doc_offsets, codes = corpus.g_doclens, corpus.g_doc_offsets, corpus.g_codes
    dev = codes.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    sigma = corpus.sigma if sigma is None else sigma
    n_pass = doclens.numel()
    targets = torch.randint(0, n_pass, (n_queries,), generator=g, device=dev)
    lens = doclens[targets].clamp(min=1)
    j = torch.arange(nq, device=dev).unsqueeze(0) % lens.unsqueeze(1)
    tok = (doc_offsets[targets].unsqueeze(1) + j).clamp(max=codes.numel() - 1)
    c = codes[tok].long()
    Q = torch.nn.functional.normalize(corpus.centroids[c] + sigma * torch.randn(n_queries, nq, corpus.dim, generator=g, device=dev), dim=-1)
    return Q.contiguous(), targets


def corpus_to_arrays(corpus, pid_base=None):
    """SyntheticCorpus (any device) -> host IndexArrays."""
    pid_base = getattr(corpus, "pid_base", 0) if pid_base is None else pid_base
    from .index import IndexArrays
    cpu = lambda t: t.detach().cpu().numpy()
    return IndexArrays(corpus.dim, corpus.nbits, cpu(corpus.codes), cpu(corpus.residuals), cpu(corpus.doclens), cpu(corpus.ivf),
                       cpu(corpus.ivf_lengths), cpu(corpus.centroids), cpu(corpus.bucket_weights),
                       bucket_cutoffs=cpu(corpus.bucket_cutoffs), pid_base=pid_base)


def corpus_device_index(corpus, pid_base=None):
    """SyntheticCorpus resident on the GPU -> DeviceIndex borrowing its tensors (no host round trip)."""
    pid_base = getattr(corpus, "pid_base", 0) if pid_base is None else pid_base
    from types import SimpleNamespace
    from .index import DeviceIndex
    meta = SimpleNamespace(dim=corpus.dim, nbits=corpus.nbits, num_centroids=corpus.K,
                           num_embeddings=int(corpus.codes.numel()), num_passages=int(corpus.doclens.numel()),
                           pid_base=pid_base, bucket_weights=corpus.bucket_weights.detach().cpu().numpy())
    tensors = {"codes": corpus.codes.contiguous(), "residuals": corpus.residuals.contiguous(),
               "doc_offsets": corpus.doc_offsets.contiguous(), "ivf_pids": corpus.ivf.contiguous(),
               "ivf_offsets": corpus.ivf_offsets.contiguous(), "centroids": corpus.centroids.contiguous()}
    return DeviceIndex(meta, device_tensors=tensors)


def make_overlapping_embeddings(n_passages, doclen, topics, seed=0, device="cuda", sub_directions=65536, chunk=1 << 22):
    """Raw token embeddings whose clusters OVERLAP -- the regime the planted-centroid corpus above hides: a token = a topic
    direction + 0.8 x a finer direction + noise (unit rows, fp16), a passage draws its tokens from three topics.  k-means has to
    find the centroids, residuals are not iid noise, and a query token is close to MANY centroids (hundreds to thousands pass
    centroid_score_threshold; fewer topics = more centroids per topic = more survivors).  Returns (embs fp16 [N, 128], doclens
    int64 [P], planted(n, nq, sigma) -> (Q fp32 [n, nq, 128], target pids)); used by profiles/built_index_probe.py, bench.py's
    built-index sub-result and tests/test_baseline_shapes.py."""
    g = torch.Generator(device=device).manual_seed(seed)
    P, L, N = n_passages, doclen, n_passages * doclen
    T = torch.nn.functional.normalize(torch.randn(topics, 128, generator=g, device=device), dim=-1)
    S = torch.nn.functional.normalize(torch.randn(sub_directions, 128, generator=g, device=device), dim=-1)
    ptop = torch.randint(0, topics, (P, 3), generator=g, device=device)
    embs = torch.empty((N, 128), dtype=torch.float16, device=device)
    for i in range(0, N, chunk):
        n = min(chunk, N - i)
        pid = torch.arange(i, i + n, device=device) // L
        top = ptop[pid, torch.randint(0, 3, (n,), generator=g, device=device)]
        sub = torch.randint(0, sub_directions, (n,), generator=g, device=device)
        v = T[top] + 0.8 * S[sub] + 0.05 * torch.randn(n, 128, generator=g, device=device)
        embs[i:i + n] = torch.nn.functional.normalize(v, dim=-1).half()
    doclens = torch.full((P,), L, dtype=torch.int64, device=device)

    def planted(n, nq=32, sigma=0.02):
        tgt = torch.randint(0, P, (n,), generator=g, device=device)
        tok = tgt.unsqueeze(1) * L + (torch.arange(nq, device=device).unsqueeze(0) % L)
        q = embs[tok.reshape(-1)].float().view(n, nq, 128)
        return torch.nn.functional.normalize(q + sigma * torch.randn(q.shape, generator=g, device=device), dim=-1).contiguous(), tgt
    return embs, doclens, planted
