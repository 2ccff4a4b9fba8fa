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

def parity_vs_reference(arrays, scorer, Q, ncells, thr, ndocs):
    """The checker's leg: every ranked list of `Q` against the reference's compiled CPU stages on the same index (oracle/_ref via
    RefCpuScorer; the C restatement where they are absent) -- tie-aware ids, scores within 1e-4 (tests/conftest.py's bar)."""
    import numpy as np
    sys.path.insert(0, os.path.join(R, "tests"))
    from conftest import tie_aware_equal
    from oracle import oracle as orc
    oi = orc.OracleIndex(arrays.dim, arrays.nbits, arrays.codes, arrays.residuals, arrays.doclens, arrays.ivf, arrays.ivf_lengths,
                         arrays.centroids, arrays.bucket_weights)
    ref = orc.RefCpuScorer(oi) if orc.ref_available() else None
    p, s, c = scorer.search_batch(Q, ndocs // 4, ncells, thr, ndocs, 32)
    scorer.check()
    p, s, c = p.cpu().numpy(), s.cpu().numpy(), c.cpu().numpy()
    Qh = Q.cpu().numpy()
    checked, worst = 0, 0.0
    for i in range(Q.size(0)):
        rp, rs, ncand = ref.rank(torch.from_numpy(Qh[i]), ncells, thr, ndocs) if ref is not None else oi.rank(Qh[i], ncells, thr, ndocs, 32)
        if ncand < ndocs:
            continue
        m = int(c[i])
        assert m == len(rp), (i, m, len(rp))
        tie_aware_equal(rp, rs, p[i, :m], s[i, :m], gap=1e-5, tol=1e-4)
        got = dict(zip(p[i, :m].tolist(), s[i, :m].tolist()))
        worst = max(worst, max(abs(got[int(a)] - float(b)) for a, b in zip(rp, rs) if int(a) in got))
        checked += 1
    return {"queries_checked": checked, "of": int(Q.size(0)), "against": "reference" if ref is not None else "port", "ranked_ids": "equal (tie-aware)",
            "max_abs_dscore": worst}


def run(P=1_000_000, L=128, nbits=2, NT=256, policies=((2, 0.45, 1024, 100), (2, 0.6, 1024, 100)), phases=True, parity_queries=0):
    N = P * L
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(1)
    t0 = time.perf_counter()
    embs, doclens, planted = synth.make_overlapping_embeddings(P, L, NT, seed=0, device=dev)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0

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
    Qs, tg = zip(*[planted(nqr) for _ in range(nb)])
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
        forms = [int(v[0]) if v.size else -1 for v in (scorer.tap(ravqa_amd._native.TAP_STAGE1_FORM, q) for q in range(256))]
        ncand = [len(scorer.tap(ravqa_amd._native.TAP_CANDIDATES, q)) for q in range(0, 256, 64)]
        out[f"search_thr{thr}"] = {"queries_per_sec": round(nqr / dt), "ms_per_step": round(dt * 1e3, 2), "recall_at_5": hit, "recall_at_100": hit100,
                                   "surviving_centroids": surv, "candidates": ncand, "stage_ms": st,
                                   "stage1_forms_of_256": {"queue": forms.count(0), "slot_untried": forms.count(1), "slot_after_queue": forms.count(2),
                                                           "small_dense": forms.count(3), "slot_after_small_dense": forms.count(4),
                                                           "dense_image": forms.count(5), "dense_exact": forms.count(6), "recompute": forms.count(7)},
                                   "index_info": scorer.device_index.info()}
        # the dominant stage against its roof (the same per-stage models as bench.py's headline: compulsory bytes / executed fp16 products)
        info = scorer.device_index.info()
        dup = info.get("duplicate_permille", 0) / 1000.0
        codes_pp = L * (1.0 - dup) if dup >= 0.1 else float(L)
        dom = max(st, key=st.get)
        P_m, ns_m = sum(ncand) / len(ncand), sum(surv) / len(surv)
        model_bytes = {"s1_filter": 4 * P_m * codes_pp + 128 * ns_m + 8 * P_m, "s2_filter_sort": 4 * ndocs * codes_pp + 16 * ndocs,
                       "s3_maxsim": (4 + 128 * nbits // 8) * (ndocs // 4) * L, "s0_centroid_scores": 4 * 128 * K / nqr}
        model_flops = {"s0_centroid_scores": 2.0 * 1.02 * K * 128 * 32, "s2_filter_sort": 2.0 * 1.1 * ndocs * codes_pp * 128 * 32,
                       "s3_maxsim": 5.0 * 2 * (ndocs // 4) * L * 128 * 32}
        roof = {"kernel": dom, "launch_ms": st[dom], "codes_per_passage": codes_pp}
        if dom in model_bytes:
            roof["hbm_GBs"] = model_bytes[dom] * nqr / (st[dom] * 1e-3) / 1e9
            roof["hbm_frac"] = roof["hbm_GBs"] / 8000.0
        if dom in model_flops:
            roof["TFLOPs"] = model_flops[dom] * nqr / (st[dom] * 1e-3) / 1e12
            roof["mfma_frac"] = roof["TFLOPs"] / 2500.0
        roof["bound"] = "mfma" if roof.get("mfma_frac", 0.0) > roof.get("hbm_frac", 0.0) else "hbm"
        roof["frac"] = max(roof.get("mfma_frac", 0.0), roof.get("hbm_frac", 0.0))
        out[f"search_thr{thr}"]["roofline"] = roof
        if parity_queries:
            out[f"search_thr{thr}"]["parity"] = parity_vs_reference(arrays, scorer, Qs[0][:parity_queries], ncells, thr, ndocs)
    del scorer, arrays
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    a = sys.argv
    print(json.dumps(run(int(a[1]) if len(a) > 1 else 1_000_000, int(a[2]) if len(a) > 2 else 128, int(a[3]) if len(a) > 3 else 2,
                         int(a[4]) if len(a) > 4 else 256)))
