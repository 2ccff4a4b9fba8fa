import sys, torch, numpy as np
sys.path.insert(0, 'profiles'); sys.path.insert(0, '.')
import ravqa_amd
from ravqa_amd import indexing, synth, _native
from ravqa_amd.scorer import IndexScorer
P, L = 1_000_000, 128
embs, doclens, planted = synth.make_overlapping_embeddings(P, L, 256, seed=0, device="cuda")
arrays = indexing.build_index(embs, doclens, nbits=2, kmeans_niters=4)
del embs
scorer = IndexScorer(arrays=arrays, max_batch=256)
Q, tgt = planted(256)
p, s, c = scorer.search_batch(Q, 100, 2, 0.45, 1024, 32)
scorer.check()
codes = torch.from_numpy(arrays.codes).cuda().view(P, L).long()
for q in (0, 64, 128):
    bits = scorer.tap(_native.TAP_IDX_BITS, q).view(np.uint32)
    surv = np.unpackbits(bits.view(np.uint8), bitorder="little").astype(bool)
    cand = torch.from_numpy(scorer.tap(_native.TAP_CANDIDATES, q).astype(np.int64)).cuda()
    sv = torch.from_numpy(surv).cuda()
    hit = sv[codes[cand]]                      # [ncand, L]
    per = hit.sum(1).float()
    cc = codes[cand]
    # distinct surviving codes per candidate
    distinct = torch.tensor([len(torch.unique(cc[i][hit[i]])) for i in range(0, cand.numel(), 97)]).float()
    print(q, "survivors", int(surv.sum()), "candidates", cand.numel(), "hits/candidate mean", float(per.mean()), "max", float(per.max()),
          "zero-hit share", float((per == 0).float().mean()), "distinct hit codes/candidate", float(distinct.mean()))
