#!/usr/bin/env python3
"""Golden vectors for the reference's CUDA-branch ARITHMETIC (FLMR_NUMERICS_GPU_FP16), produced in the build
container WITHOUT a GPU: the reference's own torch expressions of that branch, evaluated on CPU half tensors.

    python tests/golden/make_golden_gpu_numerics.py      # writes tests/golden/gpu_numerics.npz

What runs the REFERENCE's code (imported from /root/reference, CPU tensors in the dtypes its CUDA branch holds):
  * colbert_score_reduce (colbert/modeling/colbert.py:235-263): -9999 padding assigned into the half tensor, max, `.sum(-1)`;
  * StridedTensor(...).as_padded_tensor() (search/strided_tensor.py, CPU views) for every packed -> padded step;
  * the expressions of IndexScorer.score_pids' use_gpu branch (search/index_storage.py:113-149) and of
    CandidateGeneration (candidate_generation.py:12-64) re-typed below with `.cuda()` removed -- they are glue around the
    calls above: `centroids.half() @ Q.half().T`, `.max(-1).values >= thr`, `idx[codes]`, `torch.topk`;
  * F.normalize on the half tensor (residual.py:273) and `D_packed @ Q.half().T` (colbert.py:303).
What could NOT be run here and is therefore restated (recorded in the fixture's `meta.unpinned`):
  * decompress_residuals.cu (CUDA kernel): restated as half(bucket_weight) + half(centroid) in half arithmetic, which is what
    its two statements `output = weight; output += centroid` on at::Half do;
  * the CUDA GEMMs' summation order (CPU half matmul also accumulates in fp32; results can differ by one half ulp);
  * torch.topk / sort tie order on CUDA (unspecified; compared as sets / tie-aware);
  * StridedTensor's GPU-side strided views (same values as the CPU views used here).
Only DATA is written: queries, the half centroid scores, the stage survivor lists with their scores, the final ranking.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("FLMR_REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, os.path.join(REF, "third_party", "ColBERT"))
sys.path.insert(0, os.path.join(HERE, "_shims"))
os.environ.setdefault("TORCH_EXTENSIONS_DIR", "/tmp/flmr_ref_torch_ext")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import transformers  # noqa: E402

if not hasattr(transformers, "AdamW"):
    transformers.AdamW = torch.optim.AdamW

from colbert.infra.config import ColBERTConfig  # noqa: E402
from colbert.modeling.colbert import colbert_score_reduce  # noqa: E402
from colbert.search.strided_tensor import StridedTensor  # noqa: E402


def padded(t, lengths):
    """StridedTensor.as_padded_tensor on CPU (the GPU branch calls the same method on device views)."""
    return StridedTensor(t, lengths, use_gpu=False).as_padded_tensor()


def gpu_branch(z, Q, ncells, thr, ndocs, nq_cand=32):
    cfg = ColBERTConfig(total_visible_gpus=0, ncells=ncells, centroid_score_threshold=thr, ndocs=ndocs)
    centroids = torch.from_numpy(z["index.centroids_f16"]).half()            # residual.py:26: centroids.cuda().half()
    codes = torch.from_numpy(z["index.codes"])
    residuals = torch.from_numpy(z["index.residuals"])
    doclens = torch.from_numpy(z["index.doclens"])
    offsets = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(doclens, 0)])
    ivf, ivf_lengths = torch.from_numpy(z["index.ivf"]), torch.from_numpy(z["index.ivf_lengths"])
    ivf_offsets = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(ivf_lengths, 0)])
    bw = torch.from_numpy(z["index.bucket_weights"]).half()                  # residual.py:40
    rbm = torch.from_numpy(z["codec.reversed_bit_map"])
    lut = torch.from_numpy(z["codec.decompression_lookup_table"])
    Qt = torch.from_numpy(Q).unsqueeze(0)                                    # [1, Nq, 128] fp32, as dense_search passes it

    # ---- candidate_generation.py:45-64 (use_gpu) -------------------------------------------------------------------
    Qc = Qt[:, :nq_cand].squeeze(0).half()                                   # index_storage.py:77, candidate_generation.py:51
    scores = centroids @ Qc.T                                                # :13 -> half [K, nqc]
    if ncells == 1:
        cells = scores.argmax(dim=0, keepdim=True).permute(1, 0)
    else:
        cells = scores.topk(ncells, dim=0, sorted=False).indices.permute(1, 0)
    cells = cells.flatten().contiguous().unique(sorted=False)
    pids = torch.cat([ivf[ivf_offsets[c]:ivf_offsets[c + 1]] for c in cells.tolist()])
    pids = torch.unique_consecutive(pids.sort().values)

    def lookup_codes(p):
        p = p.long()
        return torch.cat([codes[offsets[i]:offsets[i + 1]] for i in p.tolist()]), doclens[p]

    # ---- index_storage.py:113-149 (use_gpu) ----------------------------------------------------------------------------
    centroid_scores = scores
    idx = centroid_scores.max(-1).values >= cfg.centroid_score_threshold
    codes_packed, codes_lengths = lookup_codes(pids)
    idx_ = idx[codes_packed.long()]
    pruned_padded, pruned_mask = padded(idx_, codes_lengths)
    pruned_lengths = (pruned_padded * pruned_mask).sum(dim=1)
    codes_packed_ = codes_packed[idx_]
    approx_ = centroid_scores[codes_packed_.long()]
    ap, am = padded(approx_, pruned_lengths)
    s1 = colbert_score_reduce(ap, am, cfg).float()
    pids1 = pids
    if cfg.ndocs < len(s1):
        pids1 = pids[torch.topk(s1, k=cfg.ndocs).indices]
    codes_packed, codes_lengths = lookup_codes(pids1)
    approx = centroid_scores[codes_packed.long()]
    ap, am = padded(approx, codes_lengths)
    s2 = colbert_score_reduce(ap, am, cfg)
    pids2 = pids1
    if cfg.ndocs // 4 < len(s2):
        pids2 = pids1[torch.topk(s2, k=cfg.ndocs // 4).indices]

    # ---- lookup_pids -> ResidualCodec.decompress (residual.py:242-278, use_gpu) with the .cu kernel restated --------------
    p = pids2.long()
    eids = torch.cat([torch.arange(int(offsets[i]), int(offsets[i + 1])) for i in p.tolist()])
    w = bw[lut[rbm[residuals[eids].long()].long()].reshape(len(eids), -1).long()]     # half(weight) per value
    D = w + centroids[codes[eids].long()]                                             # `output = w; output += centroid` in half
    D = torch.nn.functional.normalize(D, p=2, dim=-1).half()                          # residual.py:273
    lens = doclens[p]
    # ---- colbert_score_packed (colbert.py:289-311, use_gpu) -------------------------------------------------------------
    sc = D @ Qt.squeeze(0).to(dtype=D.dtype).T
    sp, sm = padded(sc, lens)
    final = colbert_score_reduce(sp, sm, cfg)
    srt = final.sort(descending=True)                                                 # index_storage.py:95
    return {"centroid_scores_f16": scores.numpy(), "cells": np.sort(cells.numpy()), "cand_pids": pids.numpy(),
            "idx": idx.numpy(), "s1_scores": s1.numpy(), "s1_pids": np.sort(pids1.numpy()),
            "s2_scores_f16": s2.numpy(), "s2_in_pids": pids1.numpy(), "s2_pids": np.sort(pids2.numpy()),
            "doc_scores_f16": final.numpy(), "doc_pids": pids2.numpy(),
            "final_pids": pids2[srt.indices].numpy(), "final_scores_f16": srt.values.numpy(),
            "D_head_f16": D[: int(lens[:4].sum())].numpy()}


def main():
    torch.set_num_threads(8)
    out = {"meta.unpinned": np.array(["decompress_residuals.cu restated as half(weight) + half(centroid)",
                                      "CUDA GEMM summation order (CPU half matmul here)",
                                      "torch.topk / sort tie order on CUDA",
                                      "StridedTensor GPU views (CPU views here)"])}
    n = 0
    for name, recs in (("idx_nb2", ["rank0", "rank3", "rank6", "rank9"]), ("idx_nb8", ["rank0", "rank2"]), ("idx_nb4", ["rank0"])):
        z = dict(np.load(os.path.join(HERE, name + ".npz")))
        for r in recs:
            ncells, thr, ndocs = int(z[f"{r}.ncells"]), float(z[f"{r}.thr"]), int(z[f"{r}.ndocs"])
            if f"{r}.undefined" in z:
                continue
            rec = gpu_branch(z, z[f"{r}.Q"], ncells, thr, ndocs, int(z[f"{r}.nq_cand"]))
            out[f"case{n}.index"] = np.array(name)
            out[f"case{n}.record"] = np.array(r)
            for k, v in rec.items():
                out[f"case{n}.{k}"] = v
            print(name, r, "cand", len(rec["cand_pids"]), "s1", len(rec["s1_pids"]), "s2", len(rec["s2_pids"]),
                  "top", rec["final_pids"][:5], rec["final_scores_f16"][:5])
            n += 1
    out["meta.n_cases"] = np.int32(n)
    path = os.path.join(HERE, "gpu_numerics.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB, {n} cases")


if __name__ == "__main__":
    main()
