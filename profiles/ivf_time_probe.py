import sys, time, torch
sys.path.insert(0, '.')
import ravqa_amd
from ravqa_amd import ops
torch.manual_seed(0)
P, L, K = 1_000_000, 128, 131072
codes = torch.randint(0, K, (P * L,), dtype=torch.int32, device="cuda")
doclens = torch.full((P,), L, dtype=torch.int64, device="cuda")
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ivf, lengths = ops.build_ivf(codes, doclens, K)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"build_ivf 1M x 128 tokens, K={K}: {dt*1e3:.1f} ms, {ivf.numel()} entries")
# spot check against torch on a slice of centroids
c = codes.long()
pid = torch.arange(P, device="cuda").repeat_interleave(L)
off = torch.zeros(K + 1, dtype=torch.int64, device="cuda"); off[1:] = torch.cumsum(lengths, 0)
for cen in (0, 1, 77777, K - 1):
    ref = torch.unique(pid[c == cen])
    got = ivf[off[cen]:off[cen + 1]].long()
    assert torch.equal(ref, got), cen
print("spot checks ok")
