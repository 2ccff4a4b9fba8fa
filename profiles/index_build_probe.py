"""Index build throughput on the GPU: HIP compress (flmr_nearest_centroids + flmr_compress_residuals) vs the torch/rocBLAS
restatement (synth.compress), tokens/s.  Usage: python profiles/index_build_probe.py [tokens] [K] [nbits]"""
import sys, time, json
import torch
sys.path.insert(0, ".")
import ravqa_amd  # noqa: F401
from ravqa_amd import synth, ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
nbits = int(sys.argv[3]) if len(sys.argv) > 3 else 2
g = torch.Generator(device="cuda").manual_seed(0)
cen = torch.nn.functional.normalize(torch.randn(K, 128, generator=g, device="cuda"), dim=-1).half().float()
codes_true = torch.randint(0, K, (n,), generator=g, device="cuda")
embs = torch.nn.functional.normalize(cen[codes_true] + 0.05 * torch.randn(n, 128, generator=g, device="cuda"), dim=-1)
cut, _ = synth.bucket_tables((embs[:100000] - cen[codes_true[:100000]]), nbits)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


t_arg, codes = timed(lambda: ops.nearest_centroids(embs, cen))
t_res, res = timed(lambda: ops.compress_residuals(embs, cen, codes, cut, nbits))


def torch_compress():
    outs = []
    for i in range(0, n, 1 << 16):  # [K, 65536] fp32 score chunks = 34 GB/s of temporaries at K = 131072
        outs.append(synth.compress(embs[i:i + (1 << 16)], cen, cut, nbits))
    return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])


t_torch, (tc, tr) = timed(torch_compress, reps=1)
agree = float((tc == codes).float().mean())
same_bytes = bool(torch.equal(ops.compress_residuals(embs, cen, tc, cut, nbits), tr))
flops = 2.0 * 128 * K * n * 2  # hi and lo chains
print(json.dumps({"tokens": n, "K": K, "nbits": nbits,
                  "hip_nearest_centroids_s": t_arg, "hip_compress_residuals_s": t_res,
                  "hip_tokens_per_s": n / (t_arg + t_res), "hip_argmax_fp16_TFLOPs": flops / t_arg / 1e12,
                  "hip_residual_GBs": n * (512 + 512 + 4 + 16 * nbits) / t_res / 1e9,
                  "torch_rocblas_s": t_torch, "torch_tokens_per_s": n / t_torch,
                  "codes_agree_with_torch_fp32": agree, "residual_bytes_equal_given_same_codes": same_bytes}))
