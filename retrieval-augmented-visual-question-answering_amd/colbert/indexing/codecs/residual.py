"""`ResidualCodec` load-side mirror (TPC/indexing/codecs/residual.py:134-150): tables + the HIP decompress op."""
from ravqa_amd import ops
from ravqa_amd.index import load_index_arrays
from ravqa_amd.scorer import _Codec


class ResidualCodec(_Codec):
    decompress_residuals = staticmethod(ops.decompress_residuals)

    @classmethod
    def load(cls, index_path, disable_gpu=False):
        return cls(load_index_arrays(index_path))
