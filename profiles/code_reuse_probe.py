"""How many DISTINCT centroid codes do the stage-1 survivors of one query hold?  (sizing of a per-query compact score table)"""
import sys, torch
sys.path.insert(0, ".")
import ravqa_amd as pkg
from ravqa_amd import synth, _native
from ravqa_amd.scorer import IndexScorer
P, DOCLEN, NB, B, NQ, k = 1_000_000, 128, 2, 8, 32, 100
K = 131072
corpus = synth.make_corpus(P, DOCLEN, K, NB, seed=0, device="cuda")
Q, _ = synth.make_queries(corpus, B, NQ, seed=2)
sc = IndexScorer(device_index=synth.corpus_device_index(corpus, pid_base=0), max_batch=B)
sc.search_batch(Q, k, 2, 0.45, 1024, 32)
for q in range(B):
    pids = torch.from_numpy(sc.tap(_native.TAP_STAGE1, q)).cuda().long()
    offs = corpus.doc_offsets
    toks = torch.cat([torch.arange(int(offs[p]), int(offs[p + 1]), device="cuda") for p in pids[:1024]])
    codes = corpus.codes[toks]
    u = torch.unique(codes).numel()
    cand = sc.tap(_native.TAP_CANDIDATES, q)
    print(f"query {q}: survivors {pids.numel()} tokens {codes.numel()} distinct codes {u} ({u / codes.numel():.3f}); candidates {cand.size}")
