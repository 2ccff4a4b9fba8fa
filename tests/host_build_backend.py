"""TEST INFRASTRUCTURE: the three device steps of `ravqa_amd.indexing.build_index` restated in torch, so that the HOST logic of
the build (sampling, the Lloyd loop, bucket tables, the on-disk format, `Indexer`'s overwrite modes) can be exercised without a
GPU.  The product module has no such branch: without `backend=` it runs libflmr_hip.so and fails loudly when there is none.
(`ravqa_amd.synth` holds the torch restatements of the codec the synthetic corpora are built with.)"""
import torch

from ravqa_amd import synth


class TorchBackend:
    @staticmethod
    def nearest_centroids(x, centroids):
        return (x.float() @ centroids.T).argmax(dim=1).to(torch.int32)       # residual.py:206-216

    @staticmethod
    def compress_residuals(x, centroids, codes, cutoffs, nbits):
        return synth.compress(x.float(), centroids, cutoffs, nbits, codes=codes)[1]   # residual.py:186-204

    @staticmethod
    def build_ivf(codes, doclens, K):
        return synth.build_ivf(codes, doclens, K)                               # indexing/utils.py:8-53
