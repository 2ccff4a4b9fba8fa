"""Index build from token embeddings, written in the reference's on-disk format so that either this
package or the reference's own `Searcher` can read it.  "Next" row 8f-1: what precedes the search path.

Restates the steps of `CollectionIndexer.run` (TPC/indexing/collection_indexer.py:56-73) that do not involve the text
encoder -- the caller supplies the (already L2-normalised) token embeddings and the per-passage token counts:

  setup    num_partitions = 2^floor(log2(16*sqrt(N)))                                   (:93)
  train    k-means on a sample (the reference calls faiss.Kmeans(dim, K, niter=kmeans_niters, seed=123), :447-463; FAISS is
           not available on ROCm here: spherical Lloyd iterations whose assignment step is the HIP argmax kernel -- same
           objective on unit vectors, different RNG, hence validated by Recall, not bit-compared), centroids L2-normalised
           (:283) and stored as half; bucket cut-offs / weights = quantiles of held-out residuals (:286-308)
  index    nearest centroid by dot product, residual, bucketize, bit-pack            (residual.py:169-204)
  finalize IVF = sorted unique pids per centroid                                    (indexing/utils.py:8-53)

Every heavy step runs on the MI355X through the C ABI (`ops`): the nearest-centroid argmax of k-means' assignment step and
of the final compression (flmr_nearest_centroids: the stage-0 fp16-split MFMA kernel with an argmax epilogue), the residual
bucketize + bit-pack (flmr_compress_residuals) and the IVF (flmr_build_ivf).  There is no host fallback in this module: the
embeddings must be on the GPU (tests of the HOST logic -- sampling, the Lloyd loop, the bucket tables, the file format -- pass
`backend=` an object with the same three functions restated in torch, tests/host_build_backend.py, exactly as they inject an
encoder).
"""
import math

import torch

from . import synth
from .index import IndexArrays


class HipBackend:
    """The three device steps of the build, on libflmr_hip.so (fails loudly without it / without a device)."""

    @staticmethod
    def nearest_centroids(x, centroids):
        from . import ops
        return ops.nearest_centroids(x, centroids)

    @staticmethod
    def compress_residuals(x, centroids, codes, cutoffs, nbits):
        from . import ops
        return ops.compress_residuals(x, centroids, codes, cutoffs, nbits)

    @staticmethod
    def build_ivf(codes, doclens, K):
        from . import ops
        return ops.build_ivf(codes, doclens, K)


def _require_unit_rows(x, what, n_check=4096):
    """The build assigns by DOT PRODUCT (k-means and the final codes): that is the reference's L2 choice (faiss.Kmeans,
    residual.py:206-220 `(centroids @ batch.T).max`) only for unit-norm rows -- what the encoders emit (colbert.py:207).  A cheap
    check on a strided sample; anything else must be normalised by the caller (silently normalising here would index different
    vectors than the ones handed in)."""
    if x.numel() == 0:
        return
    step = max(1, x.size(0) // n_check)
    nrm = x[::step][:n_check].float().norm(dim=-1)
    if bool(((nrm - 1.0).abs() > 1e-2).any()):
        raise ValueError(f"{what}: rows must be L2-normalised (norms in [{float(nrm.min()):.4f}, {float(nrm.max()):.4f}]); "
                         "the build assigns by dot product, which equals the reference's L2 assignment only on unit vectors")


def _require_device(x, backend, what):
    if backend is HipBackend and not x.is_cuda:
        raise ValueError(f"{what} must be on the GPU for the HIP build (there is no host path here; move the tensor once up "
                         "front, or pass backend= a host implementation as the tests do)")


def num_partitions_for(n_embeddings: int) -> int:
    return int(2 ** math.floor(math.log2(16 * math.sqrt(max(n_embeddings, 1)))))


def kmeans(sample, K, niters=4, seed=123, backend=HipBackend, chunk=1 << 20):
    """Spherical Lloyd iterations on unit-norm points (what the encoders emit, colbert.py:207: `normalize(D, p=2, dim=2)`):
    `niters` rounds from K distinct random sample points; assignment = nearest centroid by dot product -- for unit vectors the
    same choice as L2 -- through `backend.nearest_centroids` with the centroids re-normalised and rounded to fp16 each round
    (the index stores them as half anyway, residual.py:161; the MFMA argmax needs fp16-representable rows); empty clusters
    are re-seeded from random points.  The reference runs faiss.Kmeans (L2, its own RNG): same objective on the sphere,
    different arithmetic, hence validated by Recall, never bit-compared.  Returns fp32 centroids [K, dim] (unit rows)."""
    _require_device(sample, backend, "the k-means sample")
    _require_unit_rows(sample, "kmeans(sample)")
    g = torch.Generator(device=sample.device)
    g.manual_seed(seed)
    n = sample.size(0)
    if n < K:
        raise ValueError(f"k-means needs at least K={K} sample points, got {n}")
    sample = sample.float()
    centroids = sample[torch.randperm(n, generator=g, device=sample.device)[:K]].clone()
    for _ in range(niters):
        centroids = torch.nn.functional.normalize(centroids, dim=-1).half().float()
        assign = torch.empty(n, dtype=torch.long, device=sample.device)
        for i in range(0, n, chunk):
            assign[i:i + chunk] = backend.nearest_centroids(sample[i:i + chunk], centroids).long()
        sums = torch.zeros_like(centroids).index_add_(0, assign, sample)
        counts = torch.bincount(assign, minlength=K).to(sums.dtype)
        empty = counts == 0
        centroids = sums / counts.clamp(min=1).unsqueeze(1)
        if bool(empty.any()):
            ne = int(empty.sum())
            centroids[empty] = sample[torch.randint(0, n, (ne,), generator=g, device=sample.device)]
    return torch.nn.functional.normalize(centroids, dim=-1)


def build_index(embeddings, doclens, nbits=2, num_partitions=None, kmeans_niters=4, sample_size=None, seed=123,
                heldout_fraction=0.05, chunk=1 << 20, config=None, backend=HipBackend) -> IndexArrays:
    """embeddings: float tensor [N, 128] on the GPU, doclens: int tensor [P] with sum N.  Returns host IndexArrays
    (call `.save(path)` for the reference's directory layout)."""
    _require_device(embeddings, backend, "embeddings")
    _require_unit_rows(embeddings, "build_index(embeddings)")
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
    centroids = kmeans(train, K, kmeans_niters, seed, backend=backend).half().float()
    # ---- bucket tables from held-out residuals (:286-308) ----
    held_codes = backend.nearest_centroids(heldout, centroids).long()
    held_res = heldout - centroids[held_codes]
    cutoffs, weights = synth.bucket_tables(held_res, nbits)
    avg_residual = float(held_res.abs().mean())
    # ---- compress every embedding (residual.py:169-204) ----
    codes = torch.empty(N, dtype=torch.int32, device=dev)
    residuals = torch.empty((N, dim * nbits // 8), dtype=torch.uint8, device=dev)
    for i in range(0, N, chunk):
        e = embeddings[i:i + chunk].float()
        c = backend.nearest_centroids(e, centroids)
        codes[i:i + chunk] = c
        residuals[i:i + chunk] = backend.compress_residuals(e, centroids, c, cutoffs, nbits)
    ivf, ivf_lengths = backend.build_ivf(codes, doclens, K)
    cpu = lambda t: t.detach().cpu().numpy()
    cfg = dict(config or {})
    cfg.setdefault("kmeans_niters", kmeans_niters)
    return IndexArrays(dim, nbits, cpu(codes), cpu(residuals), cpu(doclens), cpu(ivf), cpu(ivf_lengths), cpu(centroids),
                       cpu(weights), bucket_cutoffs=cpu(cutoffs), avg_residual=avg_residual, config=cfg)
