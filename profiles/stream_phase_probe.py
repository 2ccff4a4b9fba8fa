"""Two HIP streams WITH the per-step join a stream-ordered search_batch needs: does a phase offset between the streams
(unequal first sub-batch) turn the overlap into a gain?   python profiles/stream_phase_probe.py"""
import sys, time, torch
sys.path.insert(0, ".")
from ravqa_amd import synth
from ravqa_amd.scorer import IndexScorer
P, K, B, k = 1_000_000, 131072, 1024, 100
ncells, thr, ndocs = 2, 0.45, 1024
corpus = synth.make_corpus(P, 128, K, 2, seed=0, device="cuda")
Qs = [synth.make_queries(corpus, B, 32, seed=2 + j)[0] for j in range(4)]
di = synth.corpus_device_index(corpus, pid_base=0)
def timed(fn, reps=12, warm=3):
    for i in range(warm): fn(Qs[i % 4])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(reps): fn(Qs[i % 4])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
seq = IndexScorer(device_index=di, max_batch=256, streams=1)
print("4x256 one stream:", round(timed(lambda Q: seq.search_batch(Q, k, ncells, thr, ndocs, 32)), 3))
A, Bs = IndexScorer(device_index=di, max_batch=256, streams=1), IndexScorer(device_index=di, max_batch=256, streams=1)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def sched(plan_a, plan_b, join=True):
    def run(Q):
        cur = torch.cuda.current_stream()
        if join:
            sa.wait_stream(cur); sb.wait_stream(cur)
        pos = 0
        outs = []
        for sc, st, plan in ((A, sa, plan_a), (Bs, sb, plan_b)):
            with torch.cuda.stream(st):
                for n in plan:
                    outs.append(sc.search_batch(Q[pos:pos + n], k, ncells, thr, ndocs, 32)); pos += n
        if join:
            cur.wait_stream(sa); cur.wait_stream(sb)
        return outs
    return run
for pa, pb in (((256, 256), (256, 256)), ((256, 256), (128, 256, 128)), ((256, 256), (64, 256, 192)), ((256, 256, 64), (192, 256)),
               ((256, 128), (128, 256, 256)), ((512,), (256, 256))):
    print(pa, pb, "join:", round(timed(sched(pa, pb)), 3), " free:", round(timed(sched(pa, pb, False)), 3))
