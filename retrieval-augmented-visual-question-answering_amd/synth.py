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
    if flat.numel() > 4_000_000:  # torch.quantile input limit
        flat = flat[torch.randperm(flat.numel(), device=flat.device, generator=None)[:4_000_000]]
    return flat.quantile(q[1:]), flat.quantile(q + 0.5 / n)


def build_ivf(codes, doclens, K):
    """-> (ivf pids int32 [sum unique], ivf_lengths int64 [K])."""
    n = doclens.numel()
    pid_of = torch.repeat_interleave(torch.arange(n, device=codes.device), doclens.to(codes.device))
    key = torch.unique(codes.long() * n + pid_of)
    return (key % n).to(torch.int32), torch.bincount(key // n, minlength=K).long()


class SyntheticCorpus:
    pass


def make_corpus(n_passages, doclen, K, nbits, seed=0, device="cpu", sigma=0.05, chunk_tokens=1 << 21, dim=128):
    """Clustered corpus: protos = normalize(N(0,I)[K,dim]); token = normalize(protos[c] + sigma*N(0,I)), c ~ U{0..K-1};
    the token's code is its generating centroid.  `doclen` is an int (fixed) or (lo, hi) inclusive (ragged).
    Returns a SyntheticCorpus whose tensors live on `device`."""
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
    residuals = torch.empty((N, dim * nbits // 8), dtype=torch.uint8, device=device)
    cut = wts = None
    for t0 in range(0, N, chunk_tokens):
        t1 = min(N, t0 + chunk_tokens)
        c = codes[t0:t1].long()
        emb = torch.nn.functional.normalize(centroids[c] + sigma * torch.randn(t1 - t0, dim, generator=g, device=device), dim=-1)
        if cut is None:
            cut, wts = bucket_tables((emb - centroids[c])[: 1 << 15], nbits)
        _, residuals[t0:t1] = compress(emb, centroids, cut, nbits, codes=c)
    ivf, ivf_lengths = build_ivf(codes, doclens, K)
    out = SyntheticCorpus()
    out.dim, out.nbits, out.K, out.sigma = dim, nbits, K, sigma
    out.centroids, out.doclens, out.doc_offsets = centroids, doclens, offsets
    out.codes, out.residuals = codes, residuals
    out.ivf, out.ivf_lengths = ivf, ivf_lengths
    out.ivf_offsets = torch.cat([torch.zeros(1, dtype=torch.int64, device=device), torch.cumsum(ivf_lengths, 0)])
    out.bucket_cutoffs, out.bucket_weights = cut, wts
    return out


def make_queries(corpus, n_queries, nq, seed=2, sigma=None):
    """Planted queries: query i targets passage t_i; its token j sits near the centroid of token (j mod doclen) of t_i.
    Returns (Q [n, nq, dim] on the corpus device, target pids [n])."""
    dev = corpus.codes.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    sigma = corpus.sigma if sigma is None else sigma
    n_pass = corpus.doclens.numel()
    targets = torch.randint(0, n_pass, (n_queries,), generator=g, device=dev)
    lens = corpus.doclens[targets].clamp(min=1)
    j = torch.arange(nq, device=dev).unsqueeze(0) % lens.unsqueeze(1)
    tok = (corpus.doc_offsets[targets].unsqueeze(1) + j).clamp(max=corpus.codes.numel() - 1)
    c = corpus.codes[tok].long()
    Q = torch.nn.functional.normalize(corpus.centroids[c] + sigma * torch.randn(n_queries, nq, corpus.dim, generator=g, device=dev), dim=-1)
    return Q.contiguous(), targets


def corpus_to_arrays(corpus, pid_base=0):
    """SyntheticCorpus (any device) -> host IndexArrays."""
    from .index import IndexArrays
    cpu = lambda t: t.detach().cpu().numpy()
    return IndexArrays(corpus.dim, corpus.nbits, cpu(corpus.codes), cpu(corpus.residuals), cpu(corpus.doclens), cpu(corpus.ivf),
                       cpu(corpus.ivf_lengths), cpu(corpus.centroids), cpu(corpus.bucket_weights),
                       bucket_cutoffs=cpu(corpus.bucket_cutoffs), pid_base=pid_base)


def corpus_device_index(corpus, pid_base=0):
    """SyntheticCorpus resident on the GPU -> DeviceIndex borrowing its tensors (no host round trip)."""
    from types import SimpleNamespace
    from .index import DeviceIndex
    meta = SimpleNamespace(dim=corpus.dim, nbits=corpus.nbits, num_centroids=corpus.K,
                           num_embeddings=int(corpus.codes.numel()), num_passages=int(corpus.doclens.numel()),
                           pid_base=pid_base, bucket_weights=corpus.bucket_weights.detach().cpu().numpy())
    tensors = {"codes": corpus.codes.contiguous(), "residuals": corpus.residuals.contiguous(),
               "doc_offsets": corpus.doc_offsets.contiguous(), "ivf_pids": corpus.ivf.contiguous(),
               "ivf_offsets": corpus.ivf_offsets.contiguous(), "centroids": corpus.centroids.contiguous()}
    return DeviceIndex(meta, device_tensors=tensors)
