"""An index BUILT on the device from raw token embeddings, then searched (SURVEY 8f-1 at BASELINE config 4's size):
    python profiles/built_index_probe.py [passages=1000000] [doclen=128] [nbits=2] [topics=256]
The corpus is not the planted-centroid one of bench.py: tokens overlap clusters (a token = a topic direction + a finer
direction + noise, a passage draws its tokens from three topics), so k-means has to find the centroids, residuals are not
iid noise, and a query token is close to MANY centroids -- hundreds to thousands of centroids pass centroid_score_threshold,
the regime in which stage 1 leaves the scatter form.  Prints one JSON record: build seconds per phase (k-means with the HIP
argmax as its assignment step, compression, IVF by flmr_build_ivf), queries/s + Recall@5 of planted queries at the k <= 100
policy with the per-stage split, and the surviving-centroid counts."""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import ravqa_amd
from ravqa_amd import indexing, ops, synth
from ravqa_amd.scorer import IndexScorer

def run(P=1_000_000, L=128, nbits=2, NT=256, policies=((2, 0.45, 1024, 100), (2, 0.6, 1024, 100)), phases=True):
    N = P * L
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    NS = 65536   # (NT topics: fewer topics = more centroids per topic = more survivors per query)
    T = torch.nn.functional.normalize(torch.randn(NT, 128, generator=g, device=dev), dim=-1)
    S = torch.nn.functional.normalize(torch.randn(NS, 128, generator=g, device=dev), dim=-1)
    ptop = torch.randint(0, NT, (P, 3), generator=g, device=dev)            # a passage's three topics
    t0 = time.perf_counter()
    embs = torch.empty((N, 128), dtype=torch.float16, device=dev)
    CH = 1 << 22
    for i in range(0, N, CH):
        n = min(CH, N - i)
        pid = (torch.arange(i, i + n, device=dev) // L)
        top = ptop[pid, torch.randint(0, 3, (n,), generator=g, device=dev)]
        sub = torch.randint(0, NS, (n,), generator=g, device=dev)
        v = T[top] + 0.8 * S[sub] + 0.05 * torch.randn(n, 128, generator=g, device=dev)
        embs[i:i + n] = torch.nn.functional.normalize(v, dim=-1).half()
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    doclens = torch.full((P,), L, dtype=torch.int64, device=dev)

    # ---- build, phase by phase (the same calls indexing.build_index makes; timed apart) ----
    K = indexing.num_partitions_for(N)
    t0 = time.perf_counter()
    arrays = indexing.build_index(embs, doclens, nbits=nbits, kmeans_niters=4)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    t_compress = t_ivf = None
    if phases:
        # phase split from separate calls on the built centroids (k-means = the rest)
        cen = torch.from_numpy(arrays.centroids).to(dev)
        cut = torch.from_numpy(arrays.bucket_cutoffs).to(dev)
        t0 = time.perf_counter()
        codes = torch.empty(N, dtype=torch.int32, device=dev)
        for i in range(0, N, 1 << 20):
            e = embs[i:i + (1 << 20)].float()
            c = ops.nearest_centroids(e, cen)
            codes[i:i + (1 << 20)] = c
            ops.compress_residuals(e, cen, c, cut, nbits)
        torch.cuda.synchronize()
        t_compress = time.perf_counter() - t0
        t0 = time.perf_counter()
        ivf, ivf_len = ops.build_ivf(codes, doclens, K)
        torch.cuda.synchronize()
        t_ivf = time.perf_counter() - t0
        assert torch.equal(ivf.cpu(), torch.from_numpy(arrays.ivf)) and torch.equal(codes.cpu(), torch.from_numpy(arrays.codes))

    # ---- search planted queries ----
    scorer = IndexScorer(arrays=arrays, max_batch=256)
    nqr, nb = 1024, 2
    Qs, tg = [], []
    for j in range(nb):
        tgt = torch.randint(0, P, (nqr,), generator=g, device=dev)
        tok = tgt.unsqueeze(1) * L + (torch.arange(32, device=dev).unsqueeze(0) % L)
        q = embs[tok.reshape(-1)].float().view(nqr, 32, 128)
        Qs.append(torch.nn.functional.normalize(q + 0.02 * torch.randn(q.shape, generator=g, device=dev), dim=-1).contiguous())
        tg.append(tgt)
    del embs
    out = {"passages": P, "tokens": N, "K": K, "topics": NT, "nbits": nbits, "generate_s": round(t_gen, 2), "build_index_s": round(t_build, 2),
           "ivf_entries": int(arrays.ivf.size)}
    if phases:
        out.update({"of_which_compress_s": round(t_compress, 2), "of_which_ivf_s": round(t_ivf, 3), "compress_tokens_per_s": round(N / t_compress)})
    for (ncells, thr, ndocs, k) in policies:
        for i in range(2):
            scorer.search_batch(Qs[i % nb], k, ncells, thr, ndocs, 32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(4):
            p, s, c = scorer.search_batch(Qs[i % nb], k, ncells, thr, ndocs, 32)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 4
        scorer.check()
        hit = float((p[:, :5] == tg[3 % nb].unsqueeze(1).to(torch.int32)).any(dim=1).float().mean())
        hit100 = float((p[:, :100] == tg[3 % nb].unsqueeze(1).to(torch.int32)).any(dim=1).float().mean())
        for i in range(2):
            scorer.search_batch(Qs[i % nb], k, ncells, thr, ndocs, 32, profile=True)
        st = {a: round(b / 2, 3) for a, b in scorer.stage_ms().items()}
        surv = [sum(bin(int(x)).count("1") for x in scorer.tap(ravqa_amd._native.TAP_IDX_BITS, q)) for q in range(0, 256, 32)]
        ncand = [len(scorer.tap(ravqa_amd._native.TAP_CANDIDATES, q)) for q in range(0, 256, 64)]
        out[f"search_thr{thr}"] = {"queries_per_sec": round(nqr / dt), "ms_per_step": round(dt * 1e3, 2), "recall_at_5": hit, "recall_at_100": hit100,
                                   "surviving_centroids": surv, "candidates": ncand, "stage_ms": st}
    del scorer, arrays
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    a = sys.argv
    print(json.dumps(run(int(a[1]) if len(a) > 1 else 1_000_000, int(a[2]) if len(a) > 2 else 128, int(a[3]) if len(a) > 3 else 2,
                         int(a[4]) if len(a) > 4 else 256)))
