import sys, time, torch
sys.path.insert(0, ".")
import ravqa_amd as pkg
from ravqa_amd import synth
from ravqa_amd.scorer import IndexScorer
P, DOCLEN, NB, NQ, k = 1_000_000, 128, 2, 32, 100
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ncells, thr, ndocs = 2, 0.45, 1024
K = 131072
corpus = synth.make_corpus(P, DOCLEN, K, NB, seed=0, device="cuda")
Q, _ = synth.make_queries(corpus, B, NQ, seed=2)
di = synth.corpus_device_index(corpus, pid_base=0)
def timed(fn, reps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
one = IndexScorer(device_index=di, max_batch=B)
print(f"1 stream x{B}:", timed(lambda: one.search_batch(Q, k, ncells, thr, ndocs, 32)))
for NS in (2, 4, 8):
    scs = [IndexScorer(device_index=di, max_batch=B // NS) for _ in range(NS)]
    sts = [torch.cuda.Stream() for _ in range(2)] * (NS // 2)   # NS sub-batches alternating over TWO streams
    Qs = Q.chunk(NS)
    def run():
        for sc, st, q in zip(scs, sts, Qs):
            with torch.cuda.stream(st):
                sc.search_batch(q, k, ncells, thr, ndocs, 32)
    print(NS, "streams:", timed(run))
    # same sub-batches sequentially on one stream
    def run_seq():
        for sc, q in zip(scs, Qs):
            sc.search_batch(q, k, ncells, thr, ndocs, 32)
    print(NS, "sub-batches, 1 stream:", timed(run_seq))
