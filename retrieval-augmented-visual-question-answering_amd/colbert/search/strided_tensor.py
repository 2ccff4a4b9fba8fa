"""`StridedTensor.segmented_lookup` class attribute (TPC/search/strided_tensor.py:19-37) bound to the HIP op."""
from ravqa_amd import ops
from ravqa_amd.scorer import _Strided


class StridedTensor(_Strided):
    segmented_lookup = staticmethod(ops.segmented_lookup)

    def __init__(self, packed_tensor, lengths, dim=None, use_gpu=True):
        import torch
        super().__init__(packed_tensor, torch.as_tensor(lengths))

    def lookup(self, pids, output="packed"):
        import torch
        pids = torch.as_tensor(pids).long().cpu()
        lengths, offsets = self.lengths[pids], self.offsets[pids]
        return StridedTensor.segmented_lookup(self.tensor, pids, lengths, offsets), lengths
