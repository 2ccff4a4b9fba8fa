#!/usr/bin/env python3
"""bench.py -- queries/sec (+ Recall@5) of the FLMR late-interaction search path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched with torch.distributed.run,
one rank per GPU (RCCL).  A "step" = one pass of the whole hot path (S0..S4) over one batch of `--batch` synthetic queries
(Nq=32, d=128) against the synthetic clustered corpus of BASELINE.md section 3 / SURVEY 8(d) (1 M passages x 128 tokens,
K=131072, nbits=2), index and queries already resident in HBM.  The steps ROTATE over `--query-batches` distinct batches
(4 x 1024 = 4096 queries by default), so no step re-reads the cache lines of the one before.  For N > 1 the index is
sharded by passage (BASELINE.json configs[3]); the total work is fixed, so `scaling` is "strong".  Rank 0 prints ONE JSON
line.

Besides the contract's keys the line carries (N = 1):
  roofline      the dominant kernel (largest per-stage HIP-event time inside the timed region) against its roof, from the
                kernel's OWN compulsory bytes / flops (SURVEY 8(d) counts intermediates as 0); `per_kernel` does the same
                for every stage; `whole_path` = the bytes this build must move per query against the HBM peak;
  cpu_baseline  the reference's compiled C++ stages + torch-CPU glue (oracle/_ref), >= 64 queries on all host threads and
                a second sample on 8 threads, with the ranked lists compared to the GPU's in the same run;
  sub_results   the same path at k=5, nbits=8, ragged passages and Nq=832 (short runs);
  hbm_copy_GBs  measured device copy bandwidth (read + write), next to the 8 TB/s spec used as `peak`.
N > 1: `exchange_ms` = wall time of each collective of the exact sharded protocol, measured in separate un-timed steps.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
F16_MFMA_PEAK_TFLOPS = 2500.0  # dense fp16 MFMA peak (no sparsity)
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r06", "pmc_summary.csv")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--passages", type=int, default=1_000_000)
    ap.add_argument("--doclen", type=int, default=128)
    ap.add_argument("--ragged", action="store_true", help="doclens ~ U{32..224} (mean 128) instead of a fixed length")
    ap.add_argument("--centroids", type=int, default=0, help="0 = 2^floor(log2(16*sqrt(N)))  (collection_indexer.py:93)")
    ap.add_argument("--nbits", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="queries per step (one batched _search_all_Q-style pass)")
    ap.add_argument("--sub-batch", type=int, default=256,
                    help="queries per native call: a step's batch is cut into sub-batches of this size (workspace and the "
                         "stage-2 partial buffer scale with it)")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the sub-batches of a step are dealt to")
    ap.add_argument("--query-batches", type=int, default=4, help="distinct query batches the steps rotate over")
    ap.add_argument("--nq", type=int, default=32)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--cpu-queries", type=int, default=64, help="queries timed on the CPU baseline at all threads (0 = skip)")
    ap.add_argument("--cpu-queries-8t", type=int, default=64, help="queries timed on the CPU baseline at 8 threads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-built-index", action="store_true", help="skip the built-index sub-result (~25 s: 128 M tokens indexed on the device)")
    ap.add_argument("--no-extras", action="store_true", help="skip the k=5 / nbits=8 / ragged / Nq=320 / Nq=832 sub-results")
    ap.add_argument("--replicate-stage0", action="store_true",
                    help="exact shard mode: every rank runs stage 0 for the whole batch instead of 1/N of the queries + an exchange")
    ap.add_argument("--shard-depth", type=int, default=0,
                    help="exact shard mode: sub-batches of a step in flight at once (each on its own native searcher): the exchange "
                         "of one travels while the next computes; 1 = the whole step as ONE protocol call per rank; 0 (default) = "
                         "calibrate: a few untimed steps of each, every rank takes the faster by the slowest rank's clock")
    ap.add_argument("--shard-mode", choices=["exact", "fast"], default="exact",
                    help="N > 1: exact = three key exchanges, result bit-identical to the unsharded index (default); "
                         "fast = one all-gather of per-shard top-k (superset semantics)")
    ap.add_argument("--single-device-smoke", action="store_true",
                    help="debug only: run all ranks on cuda:0 with a gloo group and a host-staged gather (exercises the "
                         "sharding / merge code on a 1-GPU box; timings are meaningless)")
    ap.add_argument("--force-distributed", action="store_true",
                    help="N = 1 only: run the sharded protocol (one shard) with its collectives issued through RCCL anyway -- "
                         "checks the torch.distributed / RCCL call path (dtypes, layouts) on a 1-GPU box")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run (one per GPU,
    127.0.0.1 rendezvous on a free port) and exit with their status.  Fewer than N visible devices is an ERROR, never a
    silent 1-GPU run (only --single-device-smoke puts several ranks on one device)."""
    import socket
    import subprocess
    if not args.single_device_smoke:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {have} HIP device(s) visible; refusing to run a smaller "
                             f"job under that name (use --single-device-smoke to exercise the N-rank control flow on one GPU)\n")
            sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this host driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def k_policy(k):
    return (2, 0.45, 1024) if k <= 100 else (4, 0.4, max(4 * k, 4096))   # searcher.py:92-118


def num_centroids(n_tok):
    import math
    return 2 ** int(math.floor(math.log2(16.0 * math.sqrt(n_tok))))       # collection_indexer.py:93


def tie_aware_same(ref_p, ref_s, got_p, gap=1e-5):
    """ids position by position, swaps allowed only inside runs of reference scores closer than `gap`."""
    if len(ref_p) != len(got_p):
        return False
    a0 = 0
    while a0 < len(ref_p):
        a1 = a0
        while a1 + 1 < len(ref_p) and abs(ref_s[a1] - ref_s[a1 + 1]) <= gap:
            a1 += 1
        if sorted(ref_p[a0:a1 + 1]) != sorted(got_p[a0:a1 + 1]):
            return False
        a0 = a1 + 1
    return True


def main():
    if os.environ.get("BENCH_WATCHDOG"):   # debugging aid: dump every thread's stack after N seconds and exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["BENCH_WATCHDOG"]), exit=True)
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)   # does not return
    import torch
    import torch.distributed as dist
    import ravqa_amd  # noqa: F401
    from ravqa_amd import _native, ops, synth
    from ravqa_amd.scorer import IndexScorer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)\n")
        sys.exit(2)
    if not args.single_device_smoke and torch.cuda.device_count() < world:
        sys.stderr.write(f"bench.py: {world} ranks but only {torch.cuda.device_count()} HIP device(s) visible\n")
        sys.exit(2)
    if args.force_distributed and world != 1:
        sys.stderr.write("bench.py: --force-distributed is an N = 1 check\n")
        sys.exit(2)
    torch.cuda.set_device(0 if args.single_device_smoke else local_rank)
    use_dist = world > 1 or args.force_distributed
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.single_device_smoke:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    doclen = (32, 224) if args.ragged else args.doclen
    K = args.centroids or num_centroids(args.passages * args.doclen)
    k = args.k
    ncells, thr, ndocs = k_policy(k)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def host_gather(t):  # --single-device-smoke only: gloo cannot gather device tensors
        parts = [torch.empty_like(t, device="cpu") for _ in range(world)]
        dist.all_gather(parts, t.cpu())
        return torch.stack(parts).cuda()

    # ---- synthetic corpus, generated on the GPU: every rank builds ITS passage shard of the same corpus (the cheap global
    # state -- centroids, doclens, one code per token -- is identical everywhere; noise, residual bytes and IVF are per shard)
    t0 = time.time()
    lo_pid, hi_pid = synth.shard_range(args.passages, rank, world)

    def build_local():
        return synth.make_corpus(args.passages, doclen, K, args.nbits, seed=0, device="cuda",
                                 pid_range=(lo_pid, hi_pid) if world > 1 else None)

    if args.single_device_smoke and world > 1:
        # all ranks share ONE device here: take turns, or eight processes' device-wide sorts starve each other
        local = None
        for r in range(world):
            if r == rank:
                local = build_local()
                torch.cuda.synchronize()
            dist.barrier()
    else:
        local = build_local()
    corpus = local   # (queries are planted over the whole corpus: make_queries reads the global doclens / codes)
    nb = max(1, args.query_batches)
    Qs, tgts = [], []
    for j in range(nb):
        Qj, tj = synth.make_queries(corpus, args.batch, args.nq, seed=2 + j)
        Qs.append(Qj)
        tgts.append(tj)
    scorer = IndexScorer(device_index=synth.corpus_device_index(local), max_batch=min(args.batch, args.sub_batch),
                         streams=args.streams)
    torch.cuda.synchronize()
    t_build = time.time() - t0

    from ravqa_amd.distributed import ShardedSearcher
    sharded = ShardedSearcher(scorer=scorer) if use_dist else None
    if sharded is not None:
        sharded.force_collectives = args.force_distributed
    exact = use_dist and args.shard_mode == "exact"

    def run_step(sc, Q, kk, pol, profile=False):
        if exact:
            # check=False: no host sync / flag exchange inside the timed steps; check_all() runs after them
            # sub-batches of the step pipelined against the exchanges (--shard-depth in flight; 1 = one call for the whole step)
            return sharded.search_batch_exact_pipelined(Q, kk, nq_cand=32, sub_batch=min(args.batch, args.sub_batch), depth=args.shard_depth,
                                                        gather=host_gather if args.single_device_smoke else None,
                                                        split_stage0=not args.replicate_stage0)
        p, s, c = sc.search_batch(Q, kk, pol[0], pol[1], pol[2], 32, profile=profile)   # query_maxlen = 32 (index_storage.py:77)
        if use_dist:
            if args.single_device_smoke:
                gs, gp = host_gather(s), host_gather(p)
            else:
                gs = torch.empty((world,) + tuple(s.shape), dtype=s.dtype, device="cuda")
                gp = torch.empty((world,) + tuple(p.shape), dtype=p.dtype, device="cuda")
                dist.all_gather_into_tensor(gs, s)
                dist.all_gather_into_tensor(gp, p)
            s, p, c = ops.merge_topk(gs, gp)
        return p, s, c

    def timed(sc, batches, targets, kk, pol, steps, warmup, collect_stages):
        for i in range(warmup):
            run_step(sc, batches[i % len(batches)], kk, pol)
        stage_sum, last = {}, {}
        barrier()
        t0_ = time.perf_counter()
        for i in range(steps):
            j = i % len(batches)
            last[j] = run_step(sc, batches[j], kk, pol, profile=collect_stages)   # HIP events on the launch streams
        barrier()
        dt = time.perf_counter() - t0_
        if collect_stages:      # one read after the timed region: the library sums the event sets of all its calls
            stage_sum = sc.stage_ms()
        if use_dist:
            t = torch.tensor([dt], device="cpu" if args.single_device_smoke else "cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        if exact:
            sharded.check_all(host_gather if args.single_device_smoke else None)   # a failed shard raises on every rank
        else:
            sc.check()
        hits = [float((last[j][0][:, :5] == targets[j].unsqueeze(1).to(torch.int32)).any(dim=1).float().mean()) for j in last]
        return dt, {n: v / steps for n, v in stage_sum.items()}, sum(hits) / len(hits), last

    shard_calibration = None
    if exact and args.shard_depth <= 0:
        # Pipelining the sub-batches hides the exchanges behind compute but runs every kernel on a quarter of the step's queries;
        # which wins depends on what the exchanges cost on this node's fabric -- measured here, before the timed region, and
        # decided identically on every rank (timed() MAX-reduces its wall time over the ranks).
        cal = {}
        for depth in (1, 2):
            args.shard_depth = depth
            cal[depth] = timed(scorer, Qs, tgts, k, (ncells, thr, ndocs), 4, 2, collect_stages=False)[0] / 4 * 1e3
        args.shard_depth = 1 if (cal[1] <= cal[2] or args.batch <= args.sub_batch) else 2
        shard_calibration = {"ms_per_step_depth1": cal[1], "ms_per_step_depth2": cal[2], "chosen_depth": args.shard_depth}
    dt, _, recall5, last = timed(scorer, Qs, tgts, k, (ncells, thr, ndocs), args.steps, args.warmup, collect_stages=False)
    # Per-stage HIP events (the roofline's kernel durations) come from a second pass over the same K steps: ten timing
    # events per native call cost 1.5 % of a step at one call per step and 6 % at four sub-batches (profiles/README.md),
    # so they stay out of the pass `value` is taken from; the instrumented pass's own step time is reported beside them.
    stage_ms, ms_per_step_events = {}, None
    if not exact and not os.environ.get('BENCH_NO_EVENTS'):
        dt_ev, stage_ms, _, _ = timed(scorer, Qs, tgts, k, (ncells, thr, ndocs), args.steps, 1, collect_stages=True)
        ms_per_step_events = dt_ev / args.steps * 1e3
    ms_per_step = dt / args.steps * 1e3
    qps = args.batch * args.steps / dt

    # ---- N > 1: wall time of every collective of the exact protocol, in separate un-timed steps ------------------------
    exchange_ms = None
    if exact:
        sharded.timings = {}   # every exchange then runs synchronously between two events on the launch stream (no host sync inside)
        nprobe = 3
        for i in range(nprobe):
            run_step(scorer, Qs[i % nb], k, (ncells, thr, ndocs))
        barrier()
        exchange_ms = {n: v / nprobe for n, v in sharded.exchange_ms().items()}
        sharded.timings = None
        # the same steps with the sub-batches NOT pipelined, for the overlap the pipeline buys (same collectives, same results)
        if args.shard_depth > 1 and args.batch > args.sub_batch:
            depth_keep, args.shard_depth = args.shard_depth, 1
            for i in range(2):
                run_step(scorer, Qs[i % nb], k, (ncells, thr, ndocs))
            barrier()
            t0_ = time.perf_counter()
            for i in range(args.steps):
                run_step(scorer, Qs[i % nb], k, (ncells, thr, ndocs))
            barrier()
            exchange_ms["ms_per_step_unpipelined"] = (time.perf_counter() - t0_) / args.steps * 1e3
            args.shard_depth = depth_keep

    # ---- N > 1, exact mode: the FAST mode beside it (north_star's collective: every shard runs S0..S4, ONE all-gather of the
    # per-shard top-k lists, merged) -- same steps, same queries, timed the same way ---------------------------------------
    fast_line = None
    if exact:
        exact = False
        try:
            dt_f, _, recall_f, _ = timed(scorer, Qs, tgts, k, (ncells, thr, ndocs), args.steps, 2, collect_stages=False)
            fast_line = {"queries_per_sec": args.batch * args.steps / dt_f, "ms_per_step": dt_f / args.steps * 1e3, "recall_at_5": recall_f,
                         "note": "--shard-mode fast on the same shards and queries: every rank runs S0..S4 over its shard at the same ndocs, one "
                                 "all-gather of the per-shard top-k (score, pid) lists, flmr_merge_topk -- exact scores over a superset of the "
                                 "unsharded survivors (the exact mode's three exchanges reproduce the unsharded result bit for bit)"}
        except Exception as e:  # noqa: BLE001
            fast_line = {"failed": repr(e)}
        exact = True

    # ---- N > 1: the REPLICA mode beside the two sharded ones -- every rank holds the WHOLE index and searches 1/N of the step's
    # queries, no collective at all: what the reference's own structure implies (one Searcher per Lightning rank, each with the
    # whole index: src/executors/FLMR_executor.py:774-798) and the throughput bound the sharded modes are to be read against
    # whenever the index fits one HBM (a 1 M-passage index is 7 GB, the 6 M one 37 GB of 288 GB); sharding is for indexes
    # larger than one HBM and for latency -------------------------------------------------------------
    replica_line = None
    if use_dist and world > 1:
        try:
            def build_whole():
                return synth.make_corpus(args.passages, doclen, K, args.nbits, seed=0, device="cuda")
            whole = None
            if args.single_device_smoke:   # (one device for every rank: take turns building)
                for r in range(world):
                    if r == rank:
                        whole = build_whole()
                        torch.cuda.synchronize()
                    dist.barrier()
            else:
                whole = build_whole()
            per = -(-args.batch // world)
            q_lo, q_hi = min(args.batch, rank * per), min(args.batch, (rank + 1) * per)
            sc_rep = IndexScorer(device_index=synth.corpus_device_index(whole), max_batch=max(1, min(per, args.sub_batch)), streams=args.streams)
            my_hits = []

            def rep_step(j):
                if q_hi > q_lo:
                    return sc_rep.search_batch(Qs[j][q_lo:q_hi], k, ncells, thr, ndocs, 32)
                return None
            for i in range(2):
                rep_step(i % nb)
            barrier()
            t0_ = time.perf_counter()
            res_ = None
            for i in range(args.steps):
                res_ = rep_step(i % nb)
            barrier()
            dt_r = time.perf_counter() - t0_
            tr_ = torch.tensor([dt_r], device="cpu" if args.single_device_smoke else "cuda", dtype=torch.float64)
            dist.all_reduce(tr_, op=dist.ReduceOp.MAX)
            dt_r = float(tr_.item())
            sc_rep.check()
            hit_ = torch.zeros(2, dtype=torch.float64, device="cpu" if args.single_device_smoke else "cuda")
            if res_ is not None:
                jl = (args.steps - 1) % nb
                hit_[0] = float((res_[0][:, :5] == tgts[jl][q_lo:q_hi].unsqueeze(1).to(torch.int32)).any(dim=1).float().sum())
                hit_[1] = q_hi - q_lo
            dist.all_reduce(hit_)
            replica_line = {"queries_per_sec": args.batch * args.steps / dt_r, "ms_per_step": dt_r / args.steps * 1e3,
                            "recall_at_5": float(hit_[0] / max(1.0, float(hit_[1]))), "queries_per_rank": per,
                            "note": "every rank holds the whole index and searches its 1/N of the step's queries: no collective in the step (the "
                                    "barrier + MAX over ranks of the contract only); results are those of the unsharded index by construction"}
            sc_rep.close_searcher()
            del sc_rep, whole
            torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001
            replica_line = {"failed": repr(e)}

    out = None
    if rank == 0:
        # ---- workload statistics of the last batch (outside the timed region) ----------------------------------------
        sub_n = min(args.batch, args.sub_batch)
        nlast = (args.batch - 1) % sub_n + 1   # taps index into the last native call: never more queries than its smallest sub-batch
        P = [len(scorer.tap(_native.TAP_CANDIDATES, q)) for q in range(0, nlast, max(1, nlast // 32))]
        ncell = [len(scorer.tap(_native.TAP_CELLS, q)) for q in range(0, nlast, max(1, nlast // 32))]
        P_mean, ncell_mean = sum(P) / len(P), sum(ncell) / len(ncell)
        d, B = 128, 128 * args.nbits // 8
        mean_len = float(local.doclens.float().mean())
        nfin_tok, ns_tok = (ndocs // 4) * mean_len, ndocs * mean_len
        ivf_mean_len = float(local.ivf_lengths.float().mean())
        nq_s0 = min(args.nq, 32)
        # the reference formulation's compulsory bytes per query (kept for comparison with round 1) ...
        alg_ref = (4 * ncell_mean * ivf_mean_len + 16 * P_mean + 4 * P_mean * mean_len + B * nfin_tok
                   + 4 * d * min(K, nfin_tok) + 4 * d * args.nq + 8 * k + 4 * d * K / args.batch)
        # ... and what THIS build must move per query: probed + surviving IVF lists (stage 1 is computed from them, the
        # candidates' codes are never read), the survivors' codes (stage 2), the finalists' codes, residual bytes and fp16
        # centroid rows (stage 3), Q, the output, and per batch: the fp32 table once for stage 0, the fp16 table once for
        # stage 2
        per_kernel_bytes = {
            "s0_centroid_scores": 4 * d * K / args.batch + 2 * 2 * d * nq_s0 + 4 * nq_s0 * K / 64,
            "s0_candidates": 2 * 4 * ncell_mean * ivf_mean_len + 8 * P_mean,
            "s2_filter_sort": 4 * ns_tok + 2 * d * K / args.batch + 16 * ndocs,
            # codes + residual bytes of the finalists' tokens, the fp16 centroid table ONCE per batch (its 256-byte rows are
            # gathered from L2 / the Infinity Cache, not from HBM), Q
            "s3_maxsim": (4 + B) * nfin_tok + 2 * d * K / args.batch + 2 * 2 * d * args.nq,
        }
        # EXECUTED MFMA flops.  Stage 0 and stage 2 run "hi first" since round 3: one fp16 product per score everywhere, the
        # second (lo) product only where a decision can depend on it -- the tiles that can hold a surviving row in stage 0
        # (< 2 % of them), the passages within the error band of the cut in stage 2 (~5 %, rescored by the gather kernel with
        # both products: counted below as 0.1 of a pass)
        per_kernel_flops = {
            "s0_centroid_scores": 2.0 * 1.02 * K * d * nq_s0,
            "s2_filter_sort": 2.0 * 1.1 * ns_tok * d * nq_s0,
            # centroid + weight form: c.q_hi, c.q_lo, w_hi.q_hi, w_hi.q_lo, w_lo.q_hi (five executed products per useful one)
            "s3_maxsim": 5.0 * 2 * nfin_tok * d * args.nq,
        }
        alg_build = sum(per_kernel_bytes.values()) + 4 * d * args.nq + 8 * k + 8 * P_mean
        if not stage_ms:  # phased multi-GPU run: per-stage events are a single-GPU measurement (see the N=1 line)
            stage_ms = {"whole_step": ms_per_step}
        pmc = {}
        try:
            import csv
            with open(PMC_SUMMARY) as f:
                for row in csv.DictReader(f):
                    pmc[row["kernel"]] = row
        except Exception:
            pass
        kname = {"s0_centroid_scores": "s0_centroid_scores", "s3_maxsim": "maxsim_lean_kernel",
                 "s2_filter_sort": "filter_stage2", "s0_candidates": "cand_fast_kernel"}

        def roof_of(stage):
            t_s = stage_ms[stage] * 1e-3
            r = {"kernel": stage, "launch_ms": stage_ms[stage]}
            if stage in per_kernel_bytes or stage == "whole_step":
                nbytes = per_kernel_bytes.get(stage, alg_build) * args.batch
                r.update({"compulsory_GB_per_launch": nbytes / 1e9, "hbm_frac": nbytes / t_s / 1e9 / HBM_PEAK_GBS})
            if stage in per_kernel_flops:
                r["TFLOPs"] = per_kernel_flops[stage] * args.batch / t_s / 1e12
                r["mfma_frac"] = r["TFLOPs"] / F16_MFMA_PEAK_TFLOPS
                # executed vs USEFUL products: S3 runs five fp16 products per useful one (the "c + w" split), stage 2 one (+ 10 %
                # for the refinement band), stage 0 one
                executed_per_useful = {"s3_maxsim": 5.0, "s2_filter_sort": 1.1}.get(stage, 1.0)
                r["useful_TFLOPs"] = r["TFLOPs"] / executed_per_useful
                r["useful_mfma_frac"] = r["useful_TFLOPs"] / F16_MFMA_PEAK_TFLOPS
            for kn, row in pmc.items():
                if stage in kname and kname[stage] in kn and args.passages == 1_000_000 and world == 1 and args.nbits == 2:
                    # the summary holds bytes per kernel LAUNCH; a step launches each kernel once per sub-batch
                    nsub = -(-args.batch // min(args.batch, args.sub_batch))
                    r["traffic"] = (2.0 * float(row["FETCH_SIZE"]) + float(row["WRITE_SIZE"])) * 1024.0 * nsub
            return r

        # ---- measured device copy bandwidth (read + write), next to the spec used as `peak` ---------------------------
        src = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
        dst = torch.empty_like(src)
        dst.copy_(src)
        torch.cuda.synchronize()
        t0_ = time.perf_counter()
        for _ in range(10):
            dst.copy_(src)
        torch.cuda.synchronize()
        copy_gbs = 10 * 2 * src.numel() * 4 / (time.perf_counter() - t0_) / 1e9
        del src, dst

        per_kernel = [roof_of(sname) for sname in sorted(stage_ms, key=stage_ms.get, reverse=True) if stage_ms[sname] > 0.05]
        dom = per_kernel[0]
        dom.setdefault("compulsory_GB_per_launch", alg_build * args.batch / 1e9)
        dom.setdefault("hbm_frac", dom["compulsory_GB_per_launch"] / (dom["launch_ms"] * 1e-3) / HBM_PEAK_GBS)
        mfma_bound = dom.get("mfma_frac", 0.0) > dom["hbm_frac"]
        roof = {"kernel": dom["kernel"], "bound": "mfma" if mfma_bound else "hbm",
                "achieved": dom["TFLOPs"] if mfma_bound else dom["compulsory_GB_per_launch"] / (dom["launch_ms"] * 1e-3),
                "peak": F16_MFMA_PEAK_TFLOPS if mfma_bound else HBM_PEAK_GBS, "unit": "TFLOP/s" if mfma_bound else "GB/s",
                "frac": dom["mfma_frac"] if mfma_bound else dom["hbm_frac"], "traffic": dom.get("traffic"),
                "peak_measured": {"hbm_copy_GBs": copy_gbs, "note": "device-to-device copy (read + write) measured in this run; the spec "
                                  "figure above is what `frac` is priced against"},
                "traffic_source": ("static: profiles/r06/pmc_summary.csv (rocprofv3 --pmc of this workload, 2*FETCH_SIZE + "
                                   "WRITE_SIZE, bytes per launch x launches per step; not measured in this run)") if dom.get("traffic") else None,
                "launch_ms": dom["launch_ms"],
                "note": ("achieved = the kernel's own compulsory bytes (or split-MFMA flops) per step / its HIP-event time per step "
                         "(summed over the step's sub-batches; events recorded on the launch stream in a second pass of the "
                         "same K steps, ms_per_step_with_stage_events); "
                         "stages 0 and 2 execute ONE fp16 product per score since round 3 (the lo product only where a decision can "
                         "depend on it), so their flops are the useful ones -- round 2's 0.26 counted two products; "
                         "stage 2 is limited by neither roof: it moves one 256-byte fp16 centroid row per survivor token "
                         "(gathered_row_GBs) -- from the Infinity Cache in the gather form (measured ceiling 9.3-9.6 TB/s), "
                         "from an L2-resident table slice per XCD in the default sliced form (69 % L2 hits, ceiling 23-32 TB/s, "
                         "profiles/microbench) whose time follows the bytes through the LDS-DMA path (DESIGN.md section 4.4)"),
                "gathered_row_GBs": (2 * d * ns_tok * args.batch / (dom["launch_ms"] * 1e-3) / 1e9) if dom["kernel"] == "s2_filter_sort" else None,
                "per_kernel": per_kernel,
                "whole_path": {"compulsory_bytes_per_query": alg_build, "GBs": alg_build * qps / 1e9,
                               "frac_of_hbm_peak": alg_build * qps / 1e9 / HBM_PEAK_GBS,
                               "reference_formulation_bytes_per_query": alg_ref}}

        out = {
            "metric": "queries/sec", "value": qps, "unit": "queries/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 (fp16-split MFMA products, fp32 accumulate; ids / keys int32 / u64)", "data": "synthetic",
            "config": {"workload": f"FLMR late-interaction search, synthetic clustered corpus {args.passages} passages x "
                                   f"{'U{32..224}' if args.ragged else args.doclen} tokens x 128-d, K={K}, nbits={args.nbits}, "
                                   f"Nq={args.nq}, k={k} (ncells={ncells}, thr={thr}, ndocs={ndocs}), {args.batch} queries/step, "
                                   f"{nb} query batches in rotation",
                       "parallelism": (f"index sharded by passage over {world} GPUs, " + (("stage 0 replicated, " if args.replicate_stage0 else "stage 0 split by query + exchange of idx bitsets/cells, ") + "all-gather of stage-1 keys + SUM all-reduces of slot-aligned stage-2/3 keys, result identical to the unsharded index" if exact else "all-gather of per-shard top-k")) if world > 1 else "1 GPU",
                       "queries_per_step": args.batch, "sub_batch": min(args.batch, args.sub_batch), "streams": args.streams},
            "recall_at_5": recall5, "roofline": roof, "cpu_baseline": None, "stage_ms_per_step": stage_ms,
            "toolchain": _native.toolchain().splitlines(),   # the compiler the loaded library was built with
            "ms_per_step_with_stage_events": ms_per_step_events,
            "candidates_per_query": P_mean, "cells_per_query": ncell_mean,
            "hbm_copy_GBs": copy_gbs, "index_build_s": t_build, "workspace_GB": scorer.workspace_bytes() / 1e9,
        }
        if shard_calibration is not None:
            out["shard_pipeline"] = shard_calibration
        if fast_line is not None:
            out["shard_mode_fast"] = fast_line
        if replica_line is not None:
            out["shard_mode_replica"] = replica_line
        if exchange_ms is not None:
            out["exchange_ms"] = exchange_ms
            out["exchange_ms_note"] = ("duration of each collective of one step between two events on the launch stream (separate "
                                       "un-timed steps, exchanges issued synchronously there); in the timed steps the sub-batches are "
                                       "pipelined -- --shard-depth of them in flight, their collectives asynchronous -- and "
                                       "ms_per_step_unpipelined is the same step as ONE protocol call per rank")

    # ---- CPU baseline (rank 0, N=1 only): the reference's own C++ stages + torch-CPU glue, bounded samples ----------------
    if rank == 0 and not use_dist and not args.no_cpu_baseline and args.cpu_queries > 0:
        try:
            from oracle import oracle as orc
            arrays = synth.corpus_to_arrays(corpus)
            oi = orc.OracleIndex(arrays.dim, arrays.nbits, arrays.codes, arrays.residuals, arrays.doclens, arrays.ivf,
                                 arrays.ivf_lengths, arrays.centroids, arrays.bucket_weights)
            p_gpu, s_gpu, _ = last[0]
            Qh = Qs[0].cpu()
            nqs = min(args.cpu_queries, args.batch)
            all_threads = torch.get_num_threads()
            if orc.ref_available():
                ref = orc.RefCpuScorer(oi, threads=0)   # 0: leave torch's thread count alone (timed at all threads, then at 8)
                ref.rank(Qh[0], ncells, thr, ndocs)   # warm
                t0_ = time.perf_counter()
                res = [ref.rank(Qh[i], ncells, thr, ndocs) for i in range(nqs)]
                tc = time.perf_counter() - t0_
                same5 = sum(res[i][0][:5] == p_gpu[i, :5].tolist() for i in range(nqs))
                samek, maxd = 0, 0.0
                for i in range(nqs):
                    rp_, rs_ = res[i][0][:k], res[i][1][:k]
                    gp_, gs_ = p_gpu[i, :k].tolist(), s_gpu[i, :k].tolist()
                    samek += int(tie_aware_same(rp_, rs_, gp_))
                    got = dict(zip(gp_, gs_))
                    maxd = max([maxd] + [abs(got[q_] - v_) for q_, v_ in zip(rp_, rs_) if q_ in got])
                n8 = min(args.cpu_queries_8t, nqs)
                qps_all = nqs / tc
                qps8 = None
                if n8 > 0:
                    torch.set_num_threads(8)
                    ref.rank(Qh[0], ncells, thr, ndocs)
                    t0_ = time.perf_counter()
                    for i in range(n8):
                        ref.rank(Qh[i], ncells, thr, ndocs)
                    qps8 = n8 / (time.perf_counter() - t0_)
                    torch.set_num_threads(all_threads)
                # the reference spawns at::get_num_threads() pthreads per extension call (filter_pids.cpp:97-101): on a
                # 128-thread host that overhead dominates, so the 8-thread run is the faster one -- report the better
                best8 = qps8 is not None and qps8 > qps_all
                # the batched OpenMP restatement SURVEY 8(d) asks for beside it: oracle/flmr_oracle.c, one query per thread,
                # all host threads, the same stages (pinned to the reference by tests/test_oracle_golden.py)
                port = None
                try:
                    ncpu = orc.effective_cpus()   # what the cgroup lets this process use (the boxes: 16 of 256 hardware threads)
                    nport = min(args.batch, max(256, 16 * ncpu))
                    oi.search_batch(Qh[:ncpu].numpy(), k, ncells, thr, ndocs, threads=ncpu)   # (warm: thread pool, page faults)
                    t0_ = time.perf_counter()
                    pp, _, _ = oi.search_batch(Qh[:nport].numpy(), k, ncells, thr, ndocs, threads=ncpu)
                    tp = time.perf_counter() - t0_
                    port = {"value": nport / tp, "unit": "queries/sec", "kind": "port", "cores": ncpu, "queries": nport,
                            "top5_identical_to_gpu": int(sum(pp[i, :5].tolist() == p_gpu[i, :5].tolist() for i in range(nport))),
                            "note": "C restatement of the same stages, OpenMP over queries (one batch call), one thread per CPU the cgroup quota allows"}
                except Exception as e:  # noqa: BLE001
                    port = {"value": None, "note": f"failed: {e!r}"}
                out["cpu_baseline"] = {
                    "value": qps8 if best8 else qps_all, "unit": "queries/sec", "cores": 8 if best8 else all_threads,
                    "kind": "reference", "cpu_model": cpu_model(), "host_threads": os.cpu_count(), "cgroup_cpus": orc.effective_cpus(),
                    "port_openmp": port,
                    "sample": (f"first {n8 if best8 else nqs} queries of batch 0 on the same 1-GPU index, one query per call (reference "
                               f"semantics); parity on the first {nqs}: top-5 ids identical to the GPU result for {same5}/{nqs}, top-{k} "
                               f"ids identical (tie-aware) for {samek}/{nqs}, max |score diff| {maxd:.2e}"),
                    "value_all_threads": qps_all, "threads_all": all_threads, "queries_all_threads": nqs,
                    "value_8_threads": qps8, "queries_8_threads": n8}
            else:
                t0_ = time.perf_counter()
                rp, _, _ = oi.search_batch(Qh[:nqs].numpy(), k, ncells, thr, ndocs)
                tc = time.perf_counter() - t0_
                same5 = sum(rp[i, :5].tolist() == p_gpu[i, :5].tolist() for i in range(nqs))
                out["cpu_baseline"] = {"value": nqs / tc, "unit": "queries/sec", "cores": os.cpu_count(), "kind": "port", "cpu_model": cpu_model(),
                                       "sample": f"first {nqs} queries of batch 0 (C restatement, OpenMP over queries); top-5 ids "
                                                 f"identical to the GPU result for {same5}/{nqs}"}
            # ---- a setting where the reference itself MISSES: noisier planted queries (sigma raised until Recall@5 of the GPU
            # path falls into 0.8..0.95), the reference's CPU path on a sample of the same queries, both recalls and the id
            # lists compared -- a wrong-but-plausible ranking cannot hide behind Recall@5 = 1.0 here
            try:
                hard = None
                for sg in (0.12, 0.16, 0.2, 0.24, 0.28, 0.32, 0.36, 0.4, 0.45, 0.5):
                    Qn, tn = synth.make_queries(corpus, 512, args.nq, seed=77, sigma=sg)
                    pn, sn, _ = scorer.search_batch(Qn, k, ncells, thr, ndocs, 32)
                    rec = float((pn[:, :5] == tn.unsqueeze(1).to(torch.int32)).any(dim=1).float().mean())
                    hard = (sg, Qn, tn, pn, sn, rec)
                    if rec <= 0.95:
                        break
                sg, Qn, tn, pn, sn, rec = hard
                nh = min(64, args.cpu_queries)
                if orc.ref_available() and nh > 0:
                    torch.set_num_threads(8)
                    Qnh = Qn.cpu()
                    resn = [ref.rank(Qnh[i], ncells, thr, ndocs) for i in range(nh)]
                    torch.set_num_threads(all_threads)
                    tl = tn[:nh].tolist()
                    rec_ref = sum(int(tl[i] in resn[i][0][:5]) for i in range(nh)) / nh
                    rec_gpu = sum(int(tl[i] in pn[i, :5].tolist()) for i in range(nh)) / nh
                    same5 = sum(resn[i][0][:5] == pn[i, :5].tolist() for i in range(nh))
                    samek = sum(int(tie_aware_same(resn[i][0][:k], resn[i][1][:k], pn[i, :k].tolist())) for i in range(nh))
                    miss_same = sum(int((tl[i] in resn[i][0][:5]) == (tl[i] in pn[i, :5].tolist())) for i in range(nh))
                    out["recall_discriminating"] = {
                        "query_sigma": sg, "recall_at_5_gpu_512_queries": rec, "queries_compared": nh,
                        "recall_at_5_reference_cpu": rec_ref, "recall_at_5_gpu_same_queries": rec_gpu,
                        "hit_or_miss_identical": miss_same, "top5_ids_identical": same5, f"top{k}_ids_identical_tie_aware": samek,
                        "note": "planted queries with token noise sigma raised (corpus sigma 0.05) until the path misses; the "
                                "reference's CPU stages (oracle/_ref) on the first queries of the same batch"}
                else:
                    out["recall_discriminating"] = {"query_sigma": sg, "recall_at_5_gpu_512_queries": rec, "queries_compared": 0}
                del Qn, tn, pn, sn
            except Exception as e:  # noqa: BLE001
                out["recall_discriminating"] = {"failed": repr(e)}
            del arrays, oi
        except Exception as e:  # the baseline must never take the bench line down
            out["cpu_baseline"] = {"value": None, "unit": "queries/sec", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}

    # ---- sub-results (N = 1): the same path at the other operating points SURVEY 8(d) names -----------------------------
    if rank == 0 and not use_dist and not args.no_extras:
        subs = []

        def sub_roofline(sc, st_, pol, nq_full, nbits_, queries_per_step):
            """The dominant stage of a sub-result against its roof, from ITS workload: candidates / surviving centroids read off the
            searcher's taps after the run, the same per-stage models as the headline's (compulsory bytes, executed fp16 products)."""
            try:
                if not st_:
                    return None
                stage = max(st_, key=st_.get)
                t_s = st_[stage] * 1e-3
                info = sc.device_index.info()
                mean_len_ = sc.arrays.num_embeddings / max(1, sc.arrays.num_passages)
                K_ = sc.arrays.num_centroids
                codes_pp = mean_len_ * (1.0 - info.get("duplicate_permille", 0) / 1000.0 if info.get("duplicate_permille", 0) >= 100 else 1.0)
                P_ = [len(sc.tap(_native.TAP_CANDIDATES, q)) for q in range(0, 32, 8)]
                ns_ = [sum(bin(int(x)).count("1") for x in sc.tap(_native.TAP_IDX_BITS, q)) for q in range(0, 32, 8)]
                P_m, ns_m = sum(P_) / len(P_), sum(ns_) / len(ns_)
                nd_ = pol[2]
                B_ = 128 * nbits_ // 8
                nqc_ = min(nq_full, 32)
                bytes_ = {"s0_centroid_scores": 4 * 128 * K_ / queries_per_step + 2 * 2 * 128 * nqc_,
                          "s1_filter": 4 * P_m * codes_pp + 128 * ns_m + 8 * P_m,                      # the candidates' (distinct) codes, the surviving rows, the keys
                          "s2_filter_sort": 4 * nd_ * codes_pp + 16 * nd_,
                          "s3_maxsim": (4 + B_) * (nd_ // 4) * mean_len_ + 2 * 2 * 128 * nq_full}
                flops_ = {"s0_centroid_scores": 2.0 * 1.02 * K_ * 128 * nqc_,
                          "s2_filter_sort": 2.0 * 1.1 * nd_ * codes_pp * 128 * nqc_,
                          "s3_maxsim": (5.0 if nq_full <= 32 else 3.0) * 2 * (nd_ // 4) * mean_len_ * 128 * nq_full}
                r = {"kernel": stage, "launch_ms": st_[stage], "candidates_per_query": P_m, "surviving_centroids_per_query": ns_m,
                     "codes_per_passage": codes_pp}
                if bytes_.get(stage):
                    r["hbm_GBs"] = bytes_[stage] * queries_per_step / t_s / 1e9
                    r["hbm_frac"] = r["hbm_GBs"] / HBM_PEAK_GBS
                if stage in flops_:
                    r["TFLOPs"] = flops_[stage] * queries_per_step / t_s / 1e12
                    r["mfma_frac"] = r["TFLOPs"] / F16_MFMA_PEAK_TFLOPS
                mf, hf = r.get("mfma_frac", 0.0), r.get("hbm_frac", 0.0)
                r.update({"bound": "mfma" if mf > hf else "hbm", "frac": max(mf, hf)})
                if stage == "s1_filter":
                    r["note"] = ("compulsory bytes (the candidates' distinct codes + the surviving rows) against HBM; the dense stage-1 kernels are VALU-issue-bound, "
                                 "not bandwidth-bound (profiles/r06/sub/pmc_*_summary.csv: SQ_INSTS_VALU x 4 clocks over 1024 SIMDs = 70 % of the launch)")
                return r
            except Exception as e:  # noqa: BLE001
                return {"failed": repr(e)}

        def sub(name, sc, batches, targets, kk, note, pol=None, nbits_=None):
            try:
                pol = pol or k_policy(kk)
                dt_, _, rec, _ = timed(sc, batches, targets, kk, pol, 6, 2, collect_stages=False)
                st_ = {}
                if not use_dist:   # the stage split of this shape, from a separate instrumented pass
                    _, st_, _, _ = timed(sc, batches, targets, kk, pol, 2, 0, collect_stages=True)
                subs.append({"name": name, "value": args.batch * 6 / dt_, "unit": "queries/sec", "ms_per_step": dt_ / 6 * 1e3,
                             "recall_at_5": rec, "stage_ms_per_step": {n_: round(v_, 3) for n_, v_ in st_.items()},
                             "roofline": sub_roofline(sc, st_, pol, int(batches[0].size(1)), nbits_ or args.nbits, args.batch), "note": note})
            except Exception as e:
                subs.append({"name": name, "value": None, "note": f"failed: {e!r}"})

        # small-batch latency: the reference's own calling pattern is ONE query per rank() call (searcher.py:73-89)
        lat = []
        try:
            import statistics
            for bsz in (1, 8, 32):
                ts = []
                for i in range(220):
                    q0 = (i * bsz) % (args.batch - bsz + 1)
                    torch.cuda.synchronize()
                    t0_ = time.perf_counter()
                    scorer.search_batch(Qs[i % nb][q0:q0 + bsz], k, ncells, thr, ndocs, 32)
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0_) * 1e3)
                ts = sorted(ts[20:])
                lat.append({"batch": bsz, "calls": len(ts), "ms_p50": statistics.median(ts), "ms_p99": ts[int(0.99 * (len(ts) - 1))],
                            "ms_min": ts[0], "queries_per_sec_p50": bsz / statistics.median(ts) * 1e3})
            scorer.check()
        except Exception as e:  # noqa: BLE001
            lat.append({"failed": repr(e)})
        out["latency"] = {"per_call": lat, "note": "wall time of one search_batch call (k=100 policy) incl. launch + device sync, "
                                                     "inputs resident on the device, 200 calls after 20 warm-up calls"}
        # the boundary handed HOST buffers: Q uploaded from pinned host memory inside the timed region, ids / scores / counts
        # copied back to pinned host memory (never `value`: the contract prices inputs resident in HBM)
        try:
            Qh = [q.cpu().pin_memory() for q in Qs]
            hp = torch.empty((args.batch, k), dtype=torch.int32).pin_memory()
            hs = torch.empty((args.batch, k), dtype=torch.float32).pin_memory()
            hc = torch.empty((args.batch,), dtype=torch.int32).pin_memory()
            nrep = 8
            for i in range(nrep + 2):
                if i == 2:
                    torch.cuda.synchronize()
                    t0_ = time.perf_counter()
                p_, s_, c_ = scorer.search_batch(Qh[i % nb].to("cuda", non_blocking=True), k, ncells, thr, ndocs, 32)
                hp.copy_(p_, non_blocking=True)
                hs.copy_(s_, non_blocking=True)
                hc.copy_(c_, non_blocking=True)
            torch.cuda.synchronize()
            dt_h = (time.perf_counter() - t0_) / nrep
            out["pcie_inclusive"] = {"queries_per_sec": args.batch / dt_h, "ms_per_step": dt_h * 1e3,
                                     "bytes_up_per_step": int(Qs[0].numel() * 4), "bytes_down_per_step": int(args.batch * (8 * k + 4)),
                                     "note": "Q from pinned host memory and results back to pinned host memory inside the timed region "
                                             "(non-blocking copies on the launch stream); reported beside `value`, never as it"}
        except Exception as e:  # noqa: BLE001
            out["pcie_inclusive"] = {"failed": repr(e)}
        # the same batches through the host side of Searcher._search_all_Q: device results -> {qid: [(pid, rank, score)] * k} (the
        # Ranking layout the executors read, searcher.py:81-89).  Python object construction, not the GPU, bounds this layer.
        try:
            from ravqa_amd.data import Ranking as _R
            from ravqa_amd.searcher import Searcher as _S
            qids = list(range(args.batch))

            def api_call(Qb):   # the host side of _search_all_Q after the policy: results -> Ranking -> todict() (FLMR_executor.py:792-794)
                # (pipelined, as _search_all_Q does it: the call returns with the kernels queued; a list waits for its sub-batch)
                rows = _S.pending_lists(scorer.search_batch_pending(Qb, k, ncells, thr, ndocs, 32))
                return _R(data=dict(zip(qids, rows))).todict()

            d_ = api_call(Qs[0])   # (warm-up: imports, allocator)
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            for i in range(4):
                d_ = api_call(Qs[i % nb])
                len(d_[qids[-1]])   # (the call returns before the device is done: the last list's length waits for the last sub-batch)
            dt_api = (time.perf_counter() - t0_) / 4
            t0_ = time.perf_counter()
            for i in range(4):   # + what the executor's loop then does with it: every (pid, rank, score) of every query unpacked
                d_ = api_call(Qs[i % nb])
                for row in d_.values():
                    for pid_, rank_, score_ in row:
                        pass
            dt_all = (time.perf_counter() - t0_) / 4
            t0_ = time.perf_counter()
            for i in range(4):   # + a caller that reads the top 5 only (Recall@5)
                d_ = api_call(Qs[i % nb])
                top5 = [row[:5] for row in d_.values()]
            dt_top5 = (time.perf_counter() - t0_) / 4
            out["api_layer"] = {"queries_per_sec": args.batch / dt_api, "ms_per_step": dt_api * 1e3, "results_per_query": len(d_[0]),
                                "queries_per_sec_reading_every_tuple": args.batch / dt_all, "queries_per_sec_reading_top5": args.batch / dt_top5,
                                "note": "search_batch + Searcher.ranking_lists + Ranking(data).todict(): what a caller of _search_all_Q holds per "
                                        "1024 queries (pipelined: every 256-query sub-batch's rows are copied to pinned host memory behind its kernels and a query's "
                                        "list waits for its sub-batch when first read; tuples are built per sub-batch when a list is iterated); the two "
                                        "other rates add the caller's own reads -- reading every tuple overlaps the device's later sub-batches"}
        except Exception as e:  # noqa: BLE001
            out["api_layer"] = {"failed": repr(e)}
        sub("k5", scorer, Qs, tgts, 5, "same index, k=5 (same pruning policy as k=100, searcher.py:92-107; 5 results returned)")
        sub("k500", scorer, Qs, tgts, 500, "same index, k=500 policy (ncells=4, thr=0.4, ndocs=4096)")
        # a threshold so low that ~9 k centroids per query pass it (> the 1024 the scatter stage 1 is sized for): those queries take
        # the code-scanning stage 1 -- the cost of leaving the tuned path, which the headline number never shows
        sub("thr0.25_code_scan", scorer, Qs[:2], tgts[:2], k, "same index, centroid_score_threshold=0.25: ~8.7 k surviving centroids per query, more than the "
            "list-scatter stage 1 takes (1024) and than fit LDS as images: the exact form of the dense stage 1 (flmr_stage1_dense.hip; until "
            "round 5 the code-scanning kernel, hence the name)", pol=(2, 0.25, 1024))
        # One searcher, the two regimes ALTERNATING step by step: the stage-1 form of a query is planned on the device from the query's own
        # statistics (cand_plan_kernel, flmr_s1_dense_modes), so no step may pay for the searcher's history -- every step, and the first
        # one after a switch in particular, within a few per cent of the same policy run homogeneously
        try:
            polA, polB = k_policy(k), (2, 0.25, 1024)

            def step_ms(pol, i):
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                scorer.search_batch(Qs[i % nb], k, pol[0], pol[1], pol[2], 32)
                torch.cuda.synchronize()
                return (time.perf_counter() - t0_) * 1e3
            import statistics as _st
            homo = {}
            for tag_, pol_ in (("planted_policy", polA), ("thr0.25", polB)):
                for i in range(3):
                    step_ms(pol_, i)
                homo[tag_] = _st.median([step_ms(pol_, i) for i in range(8)])
            seq = []
            for i in range(12):
                tag_, pol_ = (("planted_policy", polA), ("thr0.25", polB))[i % 2]
                seq.append((tag_, step_ms(pol_, i)))
            dev_ = max(abs(ms_ / homo[tag_] - 1.0) for tag_, ms_ in seq)
            subs.append({"name": "alternating_policies", "value": 2 * args.batch / (homo["planted_policy"] + homo["thr0.25"]) * 1e3, "unit": "queries/sec",
                         "homogeneous_ms_per_step": homo, "alternating_ms_per_step": [[t_, round(m_, 3)] for t_, m_ in seq],
                         "max_relative_deviation_from_homogeneous": dev_,
                         "note": "the k=100 policy and centroid_score_threshold=0.25 alternating step by step on ONE searcher (every step a switch of "
                                 "regime: list-scatter queue form <-> dense exact form), each step timed with a device sync, against the medians of "
                                 "the same policies run homogeneously: the form of a query is planned from its own statistics, not from the searcher's history"})
        except Exception as e:  # noqa: BLE001
            subs.append({"name": "alternating_policies", "value": None, "note": f"failed: {e!r}"})
        try:   # what a config with total_visible_gpus = 1 (FLMR_executor.py:784) selects: the reference's CUDA-branch arithmetic
            scf = IndexScorer(device_index=scorer.device_index, max_batch=min(args.batch, args.sub_batch), streams=args.streams,
                              numerics="gpu-fp16")
            sub("gpu_fp16_numerics", scf, Qs, tgts, k, "same index and queries in the reference's CUDA-branch arithmetic (FLMR_NUMERICS_GPU_FP16: "
                "fp16 centroid scores / embeddings, -9999 padding, fp16 sums; Q rounded to fp16, so the split kernels run their hi "
                "products only) -- parity of this mode: reference expressions on CPU half tensors, tests/golden/gpu_numerics.npz")
            scf.close_searcher()
            del scf
        except Exception as e:
            subs.append({"name": "gpu_fp16_numerics", "value": None, "note": f"failed: {e!r}"})
        # the reference's flagship query shapes: FLMR = 32 text rows + 9 ROIs x 32 visual rows (Nq = 320, src/models/retriever/FLMR.py:73-99,
        # README.md:115), PreFLMR up to 832 rows; candidate generation on the first 32 rows, stage 3 on all of them
        for name_, nq_, note_ in (("nq320", 320, "same index, FLMR-sized queries (Nq=320: 32 text + 9 x 32 visual rows; candidate generation on the first 32 tokens)"),
                                  ("nq832", 832, "same index, PreFLMR-sized queries (Nq=832, candidate generation on the first 32 tokens)")):
            try:
                Q8, t8 = zip(*[synth.make_queries(corpus, args.batch, nq_, seed=40 + j) for j in range(2)])
                sc8 = IndexScorer(device_index=scorer.device_index, max_batch=min(256, args.sub_batch), streams=args.streams)
                sub(name_, sc8, list(Q8), list(t8), k, note_)
                sc8.close_searcher()
                del Q8, t8, sc8
            except Exception as e:
                subs.append({"name": name_, "value": None, "note": f"failed: {e!r}"})
        scorer.close_searcher()
        del scorer, corpus, local
        torch.cuda.empty_cache()
        for name, nbits_, dl_, note in (("nbits8", 8, args.doclen, "nbits=8 (the FLMR configs' setting), fixed doclen"),
                                        ("ragged", args.nbits, (32, 224), "doclens ~ U{32..224}, mean 128")):
            try:
                c2 = synth.make_corpus(args.passages, dl_, K, nbits_, seed=0, device="cuda")
                Q2, t2 = zip(*[synth.make_queries(c2, args.batch, args.nq, seed=2 + j) for j in range(2)])
                sc2 = IndexScorer(device_index=synth.corpus_device_index(c2), max_batch=min(args.batch, args.sub_batch), streams=args.streams)
                sub(name, sc2, list(Q2), list(t2), k, note)
                sc2.close_searcher()
                del sc2, c2, Q2, t2
                torch.cuda.empty_cache()
            except Exception as e:
                subs.append({"name": name, "value": None, "note": f"failed: {e!r}"})
        # BASELINE configs[4] as ONE rank of its 8-GPU job sees it: passages [2.25 M, 3 M) of the 6 M x 128 corpus, K = 2^18
        # (collection_indexer.py:93), nbits = 8 -- parity at this shape: tests/test_baseline_shapes.py::test_cfg5_*
        try:
            import copy
            P5, K5 = 6_000_000, num_centroids(6_000_000 * 128)
            c5 = synth.make_corpus(P5, 128, K5, 8, seed=0, device="cuda", pid_range=synth.shard_range(P5, 3, 8))
            v5 = copy.copy(c5)
            v5.g_doclens, v5.g_doc_offsets, v5.g_codes = c5.doclens, c5.doc_offsets, c5.codes   # queries planted inside the shard
            Q5, t5 = zip(*[synth.make_queries(v5, args.batch, args.nq, seed=2 + j) for j in range(2)])
            t5 = [t + c5.pid_base for t in t5]
            sc5 = IndexScorer(device_index=synth.corpus_device_index(c5), max_batch=min(args.batch, args.sub_batch), streams=args.streams)
            sub("cfg5_shard", sc5, list(Q5), list(t5), k, f"one rank's shard of the 6 M-passage corpus: 750 k passages x 128, K={K5}, nbits=8 "
                                                          f"(stage-2 table in {sc5.device_index.info()['stage2_slices']} slices)")
            sc5.close_searcher()
            del sc5, c5, v5, Q5, t5
            torch.cuda.empty_cache()
        except Exception as e:
            subs.append({"name": "cfg5_shard", "value": None, "note": f"failed: {e!r}"})
        # An index BUILT on the device from raw embeddings whose clusters overlap (SURVEY 8f-1 at config 4's size), then searched:
        # what the planted-centroid corpus above cannot show -- k-means + compression + IVF end to end, and a stage 1 whose hit
        # passages hold many surviving centroids (profiles/built_index_probe.py; --passages-sized, 4096 topics, k <= 100 policy)
        if not args.no_built_index and args.passages == 1_000_000:
            sys.path.insert(0, os.path.join(ROOT, "profiles"))
            # 4096 topics: ~95 surviving centroids per query (the list-scatter forms of stage 1); 256 topics: ~1.5 k (the dense forms:
            # fp16 images of the score rows in LDS, the band around the cut rescored exactly -- flmr_stage1_dense.hip)
            for topics, name in ((4096, "built_index_overlapping_clusters"), (256, "built_index_dense_survivors")):
                try:
                    import built_index_probe
                    rec = built_index_probe.run(args.passages, 128, args.nbits, topics, policies=((2, 0.45, 1024, 100),), phases=False,
                                                parity_queries=0 if args.no_cpu_baseline else 16)
                    sr = rec.pop("search_thr0.45")
                    subs.append({"name": name, "value": sr["queries_per_sec"], "unit": "queries/sec",
                                 "parity": sr.get("parity"),
                                 "ms_per_step": sr["ms_per_step"], "recall_at_5": sr["recall_at_5"], "stage_ms_per_step": sr["stage_ms"],
                                 "surviving_centroids_per_query": sr["surviving_centroids"], "candidates_per_query": sr["candidates"],
                                 "stage1_forms_of_256": sr.get("stage1_forms_of_256"), "index_info": sr.get("index_info"), "roofline": sr.get("roofline"), "index_build": rec,
                                 "note": f"1 M passages x 128 raw token embeddings ({topics} topics; a token = a topic direction + a finer direction + "
                                         "noise, three topics per passage) indexed end to end on the device by indexing.build_index (k-means with "
                                         "the HIP argmax as its assignment step, compression, IVF by flmr_build_ivf), then searched with planted queries"})
                except Exception as e:  # noqa: BLE001
                    subs.append({"name": name, "value": None, "note": f"failed: {e!r}"})
                torch.cuda.empty_cache()
        out["sub_results"] = subs

    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
