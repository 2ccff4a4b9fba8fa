import torch, time
x = torch.empty(1<<30, dtype=torch.float32, device="cuda")  # 4 GiB
y = torch.empty_like(x)
def t(f, n=10):
    f(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n
b = x.numel()*4
print("fill   GB/s", b/t(lambda: x.fill_(1.0))/1e9)
print("copy   GB/s (r+w)", 2*b/t(lambda: y.copy_(x))/1e9)
print("read   GB/s (sum)", b/t(lambda: x.sum())/1e9)
