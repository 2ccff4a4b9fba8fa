#!/usr/bin/env python3
"""bench.py -- queries/sec (+ Recall@5) of the FLMR late-interaction search path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched with
torch.distributed.run, one rank per GPU (RCCL).  A "step" = one pass of the whole hot path (S0..S4) over one batch of
`--batch` synthetic queries (Nq=32, d=128) against the synthetic clustered corpus of BASELINE.md section 3
(1 M passages x 128 tokens, K=131072, nbits=2), index and queries already resident in HBM.  For N > 1 the index is
sharded by passage (BASELINE.json configs[3]): every rank searches its shard for every query, then ONE all-gather of
the per-shard top-k over xGMI and a merge; the total work is fixed, so `scaling` is "strong".
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
FP32_MFMA_PEAK_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--passages", type=int, default=1_000_000)
    ap.add_argument("--doclen", type=int, default=128)
    ap.add_argument("--centroids", type=int, default=0, help="0 = 2^floor(log2(16*sqrt(N)))  (collection_indexer.py:93)")
    ap.add_argument("--nbits", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="queries per step (one batched _search_all_Q-style pass)")
    ap.add_argument("--nq", type=int, default=32)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--cpu-queries", type=int, default=12, help="queries of the batch timed on the CPU baseline (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--replicate-stage0", action="store_true",
                    help="exact shard mode: every rank runs stage 0 for the whole batch instead of 1/N of the queries + an exchange")
    ap.add_argument("--shard-mode", choices=["exact", "fast"], default="exact",
                    help="N > 1: exact = three key exchanges, result bit-identical to the unsharded index (default); "
                         "fast = one all-gather of per-shard top-k (superset semantics)")
    ap.add_argument("--single-device-smoke", action="store_true",
                    help="debug only: run all ranks on cuda:0 with a gloo group and a host-staged gather (exercises the "
                         "sharding / merge code on a 1-GPU box; timings are meaningless)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import ravqa_amd
    from ravqa_amd import synth, ops
    from ravqa_amd.scorer import IndexScorer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, (world, args.gpus)
    torch.cuda.set_device(0 if args.single_device_smoke else local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.single_device_smoke:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    N_tok = args.passages * args.doclen
    K = args.centroids or 2 ** int(torch.log2(torch.tensor(16.0 * (N_tok ** 0.5))).floor())
    k = args.k
    ncells, thr, ndocs = (2, 0.45, 1024) if k <= 100 else (4, 0.4, max(4 * k, 4096))   # searcher.py:92-118

    # ---- synthetic corpus, generated on the GPU, identical on every rank; then this rank's passage shard --------
    t0 = time.time()
    corpus = synth.make_corpus(args.passages, args.doclen, K, args.nbits, seed=0, device="cuda")
    Q, targets = synth.make_queries(corpus, args.batch, args.nq, seed=2)
    if world > 1:
        lo, hi = (args.passages * rank) // world, (args.passages * (rank + 1)) // world
        tlo, thi = int(corpus.doc_offsets[lo]), int(corpus.doc_offsets[hi])
        keep = (corpus.ivf >= lo) & (corpus.ivf < hi)
        owner = torch.repeat_interleave(torch.arange(K, device="cuda"), corpus.ivf_lengths)
        shard = synth.SyntheticCorpus()
        shard.dim, shard.nbits, shard.K, shard.sigma = corpus.dim, corpus.nbits, K, corpus.sigma
        shard.centroids, shard.bucket_weights, shard.bucket_cutoffs = corpus.centroids, corpus.bucket_weights, corpus.bucket_cutoffs
        shard.codes, shard.residuals = corpus.codes[tlo:thi].contiguous(), corpus.residuals[tlo:thi].contiguous()
        shard.doclens = corpus.doclens[lo:hi].contiguous()
        shard.doc_offsets = (corpus.doc_offsets[lo:hi + 1] - tlo).contiguous()
        shard.ivf = (corpus.ivf[keep] - lo).to(torch.int32).contiguous()
        shard.ivf_lengths = torch.bincount(owner[keep], minlength=K).long()
        shard.ivf_offsets = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), torch.cumsum(shard.ivf_lengths, 0)])
        local, pid_base = shard, lo
    else:
        local, pid_base = corpus, 0
    scorer = IndexScorer(device_index=synth.corpus_device_index(local, pid_base=pid_base), max_batch=args.batch)
    torch.cuda.synchronize()
    t_build = time.time() - t0

    from ravqa_amd.distributed import ShardedSearcher
    sharded = ShardedSearcher(scorer=scorer) if world > 1 else None

    def host_gather(t):  # --single-device-smoke only: gloo cannot gather device tensors
        parts = [torch.empty_like(t, device="cpu") for _ in range(world)]
        dist.all_gather(parts, t.cpu())
        return torch.stack(parts).cuda()

    def step(profile=False):
        if world > 1 and args.shard_mode == "exact":
            return sharded.search_batch_exact(Q, k, nq_cand=32, gather=host_gather if args.single_device_smoke else None,
                                              split_stage0=not args.replicate_stage0)
        p, s, c = scorer.search_batch(Q, k, ncells, thr, ndocs, 32, profile=profile)  # query_maxlen = 32 (index_storage.py:77)
        if world > 1:
            gs = torch.empty((world,) + tuple(s.shape), dtype=s.dtype, device="cuda")
            gp = torch.empty((world,) + tuple(p.shape), dtype=p.dtype, device="cuda")
            if args.single_device_smoke:
                hs, hp = [torch.empty_like(s, device="cpu") for _ in range(world)], [torch.empty_like(p, device="cpu") for _ in range(world)]
                dist.all_gather(hs, s.cpu())
                dist.all_gather(hp, p.cpu())
                gs, gp = torch.stack(hs).cuda(), torch.stack(hp).cuda()
            else:
                dist.all_gather_into_tensor(gs, s)
                dist.all_gather_into_tensor(gp, p)
            s, p, c = ops.merge_topk(gs, gp)
        return p, s, c

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    stage_sum = {}
    barrier()
    t0 = time.perf_counter()
    staged = world > 1 and args.shard_mode == "exact"   # the phased protocol has no per-stage event set
    for _ in range(args.steps):
        p, s, c = step(profile=True)
        if not staged:
            for name, ms in scorer.stage_ms().items():   # HIP events on the launch stream, recorded inside the timed region
                stage_sum[name] = stage_sum.get(name, 0.0) + ms
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cpu" if args.single_device_smoke else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    qps = args.batch * args.steps / dt
    stage_ms = {n: v / args.steps for n, v in stage_sum.items()}

    # ---- workload statistics of the last batch (outside the timed region) --------------------------------------------
    from ravqa_amd import _native
    P = [len(scorer.tap(_native.TAP_CANDIDATES, q)) for q in range(0, args.batch, max(1, args.batch // 32))]
    ncell = [len(scorer.tap(_native.TAP_CELLS, q)) for q in range(0, args.batch, max(1, args.batch // 32))]
    P_mean, ncell_mean = sum(P) / len(P), sum(ncell) / len(ncell)
    recall5 = float((p[:, :5] == targets.unsqueeze(1).to(torch.int32)).any(dim=1).float().mean())
    d, B = 128, 128 * args.nbits // 8
    nfin_tok = (ndocs // 4) * args.doclen
    ivf_mean_len = float(local.ivf_lengths.float().mean())
    # SURVEY 8(d) algorithmic bytes per query (per shard): IVF lists + (doclen,offset) + S1 code scan + residuals of
    # the finalists + their centroid rows + Q + output + centroid matrix amortised over the batch
    alg_bytes = (4 * ncell_mean * ivf_mean_len + 16 * P_mean + 4 * P_mean * args.doclen + B * nfin_tok
                 + 4 * d * min(K, nfin_tok) + 4 * d * args.nq + 8 * k + 4 * d * K / args.batch)
    s1_bytes = 16 * P_mean + 4 * P_mean * args.doclen
    s0_flops = 2.0 * K * d * min(args.nq, 32)
    if not stage_ms:  # phased multi-GPU run: per-stage events are a single-GPU measurement (see the N=1 line)
        stage_ms = {"whole_step": ms_per_step}
    dom = max(stage_ms, key=stage_ms.get)
    dom_ms_per_query = stage_ms[dom] / args.batch
    if dom == "s0_centroid_scores":
        ach = s0_flops / (dom_ms_per_query * 1e-3) / 1e12
        roof = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": None}
    else:
        nbytes = {"s1_filter": s1_bytes, "s3_maxsim": B * nfin_tok + 4 * d * min(K, nfin_tok) + 4 * nfin_tok,
                  "s2_filter_sort": 4 * ndocs * args.doclen + 4 * 32 * ndocs * args.doclen}.get(dom, alg_bytes)
        ach = nbytes / (dom_ms_per_query * 1e-3) / 1e9
        roof = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": None}
    roof["launch_ms"] = stage_ms[dom]
    # HBM bytes per launch of the dominant kernel from the PMC passes (profiles/pmc_passes.sh: FETCH_SIZE doubled as the
    # gfx950 guide prescribes, + WRITE_SIZE), when a summary for this round has been committed next to this file
    try:
        import csv
        kname = {"s1_filter": "filter_stage1_kernel", "s0_centroid_scores": "s0_centroid_scores_f16", "s3_maxsim": "maxsim_f16_kernel",
                 "s2_filter_sort": "filter_stage2", "s0_candidates": "cand_mark_score_kernel"}.get(dom)
        with open(os.path.join(ROOT, "profiles", "pmc_summary_latest.csv")) as f:
            for row in csv.DictReader(f):
                if kname and kname in row["kernel"] and args.passages == 1_000_000 and world == 1:
                    roof["traffic"] = (2.0 * float(row["FETCH_SIZE"]) + float(row["WRITE_SIZE"])) * 1024.0
                    roof["traffic_unit"] = "bytes per launch (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc, profiles/)"
    except Exception:
        pass
    roof["whole_path_algorithmic_GBs"] = alg_bytes * qps / 1e9
    roof["whole_path_frac_of_hbm_peak"] = alg_bytes * qps / 1e9 / HBM_PEAK_GBS

    # ---- CPU baseline (rank 0, N=1 only): the reference's own C++ stages + torch-CPU glue, bounded sample ----------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.cpu_queries > 0:
        try:
            from oracle import oracle as orc
            arrays = synth.corpus_to_arrays(corpus)
            oi = orc.OracleIndex(arrays.dim, arrays.nbits, arrays.codes, arrays.residuals, arrays.doclens, arrays.ivf,
                                 arrays.ivf_lengths, arrays.centroids, arrays.bucket_weights)
            nqs = min(args.cpu_queries, args.batch)
            Qh = Q[:nqs].cpu()
            same5 = 0
            parity_note = ""
            if orc.ref_available():
                ref = orc.RefCpuScorer(oi)
                ref.rank(Qh[0], ncells, thr, ndocs)   # warm
                t0 = time.perf_counter()
                res = [ref.rank(Qh[i], ncells, thr, ndocs) for i in range(nqs)]
                tc = time.perf_counter() - t0
                kind, cores = "reference", torch.get_num_threads()
                same5 = sum(res[i][0][:5] == p[i, :5].tolist() for i in range(nqs))
                # full top-k against the reference: ids position by position, swaps allowed only inside runs of reference
                # scores closer than 1e-5 (another valid fp32 summation order may swap those); scores by pid
                samek, maxd = 0, 0.0
                for i in range(nqs):
                    rp_, rs_ = res[i][0][:k], res[i][1][:k]
                    gp_, gs_ = p[i, :k].tolist(), s[i, :k].tolist()
                    got = dict(zip(gp_, gs_))
                    ok, a0 = len(rp_) == len(gp_), 0
                    while ok and a0 < len(rp_):
                        a1 = a0
                        while a1 + 1 < len(rp_) and abs(rs_[a1] - rs_[a1 + 1]) <= 1e-5:
                            a1 += 1
                        ok = sorted(rp_[a0:a1 + 1]) == sorted(gp_[a0:a1 + 1])
                        a0 = a1 + 1
                    samek += int(ok)
                    maxd = max([maxd] + [abs(got[q_] - v_) for q_, v_ in zip(rp_, rs_) if q_ in got])
                parity_note = f"; top-{k} ids identical (tie-aware) for {samek}/{nqs}, max |score diff| {maxd:.2e}"
            else:
                t0 = time.perf_counter()
                rp, _, _ = oi.search_batch(Qh.numpy(), k, ncells, thr, ndocs)
                tc = time.perf_counter() - t0
                kind, cores = "port", os.cpu_count()
                same5 = sum(rp[i, :5].tolist() == p[i, :5].tolist() for i in range(nqs))
            cpu = {"value": nqs / tc, "unit": "queries/sec", "cores": cores, "kind": kind,
                   "sample": f"first {nqs} queries of the same batch on the same 1-GPU index, one query per call "
                             f"(reference semantics); top-5 ids identical to the GPU result for {same5}/{nqs}" + parity_note}
        except Exception as e:  # the baseline must never take the bench line down
            cpu = {"value": None, "unit": "queries/sec", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}

    if rank == 0:
        out = {
            "metric": "queries/sec", "value": qps, "unit": "queries/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"FLMR late-interaction search, synthetic clustered corpus {args.passages} passages x "
                                   f"{args.doclen} tokens x 128-d, K={K}, nbits={args.nbits}, Nq={args.nq}, k={k} "
                                   f"(ncells={ncells}, thr={thr}, ndocs={ndocs}), {args.batch} queries/step",
                       "parallelism": (f"index sharded by passage over {world} GPUs, " + (("stage 0 replicated, " if args.replicate_stage0 else "stage 0 split by query + exchange of idx bitsets/cells, ") + "all-gather of stage-1 keys + SUM all-reduces of slot-aligned stage-2/3 keys, result identical to the unsharded index" if args.shard_mode == "exact" else "all-gather of per-shard top-k")) if world > 1 else "1 GPU",
                       "queries_per_step": args.batch},
            "recall_at_5": recall5,
            "roofline": roof,
            "cpu_baseline": cpu,
            "stage_ms_per_step": stage_ms,
            "candidates_per_query": P_mean, "cells_per_query": ncell_mean,
            "algorithmic_bytes_per_query": alg_bytes,
            "index_build_s": t_build, "workspace_GB": scorer.workspace_bytes() / 1e9,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
