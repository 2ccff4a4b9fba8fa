"""Index build from token embeddings, written in the reference's on-disk format so that either this
package or the reference's own `Searcher` can read it.  "Next" row 8f-1: what precedes the search path.

Restates the steps of `CollectionIndexer.run` (TPC/indexing/collection_indexer.py:56-73) that do not involve the text
encoder -- the caller supplies the (already L2-normalised) token embeddings and the per-passage token counts:

  setup    num_partitions = 2^floor(log2(16*sqrt(N)))                                   (:93)
  train    k-means on a sample (the reference calls faiss.Kmeans(dim, K, niter=kmeans_niters, seed=123), :447-463; FAISS is
           not available on ROCm here, so this is a plain Lloyd iteration with L2 assignment in torch -- same objective,
           different RNG, hence validated by Recall, not bit-compared), centroids L2-normalised (:283) and stored as half;
           bucket cut-offs / weights = quantiles of held-out residuals (:286-308)
  index    nearest centroid by dot product, residual, bucketize, bit-pack            (synth.compress = residual.py:169-204)
  finalize IVF = sorted unique pids per centroid                                    (synth.build_ivf = indexing/utils.py:8-53)

With the embeddings on the GPU the `index` step runs on the HIP kernels of csrc/flmr_build.hip (ops.compress); k-means and
the quantiles use torch on the same device.
"""
import math

import torch

from . import synth
from .index import IndexArrays


def num_partitions_for(n_embeddings: int) -> int:
    return int(2 ** math.floor(math.log2(16 * math.sqrt(max(n_embeddings, 1)))))


def _assign_l2(x, centroids, chunk=1 << 16):
    """argmin_c ||x - c||^2 = argmax_c (x.c - |c|^2/2), chunked over x."""
    half_sq = 0.5 * (centroids * centroids).sum(-1)
    out = torch.empty(x.size(0), dtype=torch.long, device=x.device)
    for i in range(0, x.size(0), chunk):
        out[i:i + chunk] = (x[i:i + chunk] @ centroids.T - half_sq).argmax(dim=1)
    return out


def kmeans(sample, K, niters=4, seed=123):
    """Lloyd's algorithm (L2), `niters` iterations from K distinct random sample points; empty clusters are re-seeded from
    random points.  Returns fp32 centroids [K, dim]."""
    g = torch.Generator(device=sample.device)
    g.manual_seed(seed)
    n = sample.size(0)
    if n < K:
        raise ValueError(f"k-means needs at least K={K} sample points, got {n}")
    centroids = sample[torch.randperm(n, generator=g, device=sample.device)[:K]].clone().float()
    for _ in range(niters):
        assign = _assign_l2(sample, centroids)
        sums = torch.zeros_like(centroids).index_add_(0, assign, sample.float())
        counts = torch.bincount(assign, minlength=K).to(sums.dtype)
        empty = counts == 0
        centroids = sums / counts.clamp(min=1).unsqueeze(1)
        if bool(empty.any()):
            ne = int(empty.sum())
            centroids[empty] = sample[torch.randint(0, n, (ne,), generator=g, device=sample.device)].float()
    return centroids


def build_index(embeddings, doclens, nbits=2, num_partitions=None, kmeans_niters=4, sample_size=None, seed=123,
                heldout_fraction=0.05, chunk=1 << 20, config=None) -> IndexArrays:
    """embeddings: float tensor [N, 128] (any device), doclens: int tensor [P] with sum N.  Returns host IndexArrays
    (call `.save(path)` for the reference's directory layout)."""
    dev = embeddings.device
    N, dim = embeddings.shape
    doclens = torch.as_tensor(doclens, device=dev).long()
    assert int(doclens.sum()) == N, "doclens must sum to the number of embeddings"
    K = num_partitions or num_partitions_for(N)
    K = min(K, N)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    # ---- sample + held-out split (collection_indexer.py:206-256) ----
    sample_size = min(N, sample_size or max(K * 40, 1 << 14))
    perm = torch.randperm(N, generator=g, device=dev)[:sample_size]
    sample = embeddings[perm].float()
    n_held = max(1, min(int(heldout_fraction * sample_size), 50_000))
    heldout, train = sample[:n_held], sample[n_held:] if sample_size - n_held >= K else sample
    centroids = torch.nn.functional.normalize(kmeans(train, K, kmeans_niters, seed), dim=-1).half().float()
    # ---- bucket tables from held-out residuals (:286-308) ----
    held_codes = (centroids @ heldout.T).argmax(dim=0)
    held_res = heldout - centroids[held_codes]
    cutoffs, weights = synth.bucket_tables(held_res, nbits)
    avg_residual = float(held_res.abs().mean())
    # ---- compress every embedding (residual.py:169-204) ----
    codes = torch.empty(N, dtype=torch.int32, device=dev)
    residuals = torch.empty((N, dim * nbits // 8), dtype=torch.uint8, device=dev)
    if dev.type == "cuda":
        from . import ops  # HIP kernels: fp16-split MFMA argmax + fused residual/bucketize/bit-pack (csrc/flmr_build.hip)
        compress = lambda e: ops.compress(e, centroids, cutoffs, nbits)
    else:  # host tensors (tests, tiny corpora): the torch restatement of the same steps
        compress = lambda e: synth.compress(e, centroids, cutoffs, nbits)
    for i in range(0, N, chunk):
        c, r = compress(embeddings[i:i + chunk].float())
        codes[i:i + chunk], residuals[i:i + chunk] = c, r
    ivf, ivf_lengths = synth.build_ivf(codes, doclens, K)
    cpu = lambda t: t.detach().cpu().numpy()
    cfg = dict(config or {})
    cfg.setdefault("kmeans_niters", kmeans_niters)
    return IndexArrays(dim, nbits, cpu(codes), cpu(residuals), cpu(doclens), cpu(ivf), cpu(ivf_lengths), cpu(centroids),
                       cpu(weights), bucket_cutoffs=cpu(cutoffs), avg_residual=avg_residual, config=cfg)
