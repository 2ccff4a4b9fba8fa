"""Can two stages of DIFFERENT sub-batches share a CU?  Stage 2's workgroups request 78 KB of LDS, two fill a CU (156 of 160 KB),
so while stage 2 runs nothing else is resident beside it; with 81 KB per workgroup (build flag -DX2_LDS_PAD=3072) only ONE fits
and a stage-3 workgroup (77 KB) of another sub-batch can sit next to it.  Stage 2 alone loses 28 % that way;
this probe measures what two free-running streams (one IndexScorer each, alternate sub-batches of 256, no join between steps)
make of it, per library variant:
    FLMR_HIP_LIB=.../libflmr_hip_x2pad.so python profiles/corun_probe.py
"""
import os, sys, time, torch
sys.path.insert(0, ".")
from ravqa_amd import synth
from ravqa_amd.scorer import IndexScorer

P, K, k = 1_000_000, 131072, 100
ncells, thr, ndocs = 2, 0.45, 1024
corpus = synth.make_corpus(P, 128, K, 2, seed=0, device="cuda")
Qs = [synth.make_queries(corpus, 1024, 32, seed=2 + j)[0] for j in range(2)]
di = synth.corpus_device_index(corpus, pid_base=0)
tag = os.path.basename(os.environ.get("FLMR_HIP_LIB", "product library"))
SUB = int(os.environ.get("CORUN_SUB", "256"))

seq = IndexScorer(device_index=di, max_batch=SUB, streams=1)
def one_stream(reps=10, warm=3):
    for i in range(warm): seq.search_batch(Qs[i % 2], k, ncells, thr, ndocs, 32)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(reps): seq.search_batch(Qs[i % 2], k, ncells, thr, ndocs, 32)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

def free_running(nstreams, reps=10, warm=3, stagger=True):
    scs = [IndexScorer(device_index=di, max_batch=SUB, streams=1) for _ in range(nstreams)]
    sts = [torch.cuda.Stream() for _ in range(nstreams)]
    per = 1024 // SUB   # sub-batches per 1024 queries
    def run(n_steps):
        outs = []
        for step in range(n_steps):
            Q = Qs[step % 2]
            for j in range(per):
                s = j % nstreams
                with torch.cuda.stream(sts[s]):
                    outs.append(scs[s].search_batch(Q[j * SUB:(j + 1) * SUB], k, ncells, thr, ndocs, 32))
        return outs
    if stagger:   # start the streams out of phase: stream s first runs a short call of s * SUB / nstreams queries
        for s in range(1, nstreams):
            with torch.cuda.stream(sts[s]):
                scs[s].search_batch(Qs[0][:max(8, s * SUB // nstreams)], k, ncells, thr, ndocs, 32)
    run(warm)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = run(reps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e3
    for sc in scs: sc.check()
    # same results as the one-stream scorer (the last step's first sub-batch)
    ref = seq.search_batch(Qs[(reps - 1) % 2][:SUB], k, ncells, thr, ndocs, 32)
    same = bool((ref[0] == outs[-per][0]).all()) and bool((ref[1] == outs[-per][1]).all())
    return dt, same

print(f"[{tag}] sub-batch {SUB}")
print(f"  one stream, {1024 // SUB} x {SUB} in sequence : {one_stream():7.3f} ms per 1024 queries")
for ns in (2, 3):
    dt, same = free_running(ns)
    print(f"  {ns} free-running streams              : {dt:7.3f} ms per 1024 queries   (results identical: {same})")
print(f"  one stream again                      : {one_stream():7.3f} ms per 1024 queries")
