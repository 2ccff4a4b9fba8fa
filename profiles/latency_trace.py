"""Single-query calls under `rocprofv3 --kernel-trace`: what one search_batch call of 1 (or N) queries consists of on the device --
kernels, their durations and the gaps between them (launch-bound or not?).
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $REPO/profiles/latency_trace.py [batch]
    python $REPO/profiles/latency_trace.py --report $OUT/t_kernel_trace.csv"""
import csv
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def report(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    # the last call: walk back from the last sort_topn_kernel to the s0_split_q before it
    last = max(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("sort_topn_kernel"))
    first = max(i for i, r in enumerate(rows[:last]) if "s0_split_q" in r["Kernel_Name"])
    call = rows[first:last + 1]
    t0 = int(call[0]["Start_Timestamp"])
    busy, prev_end = 0, None
    for r in call:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = 0 if prev_end is None else s - prev_end
        busy += e - s
        print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:6.1f}  gap {gap / 1e3:5.1f}  {r['Kernel_Name'][:60]}")
        prev_end = e
    span = int(call[-1]["End_Timestamp"]) - t0
    print(f"{len(call)} kernels, span {span / 1e3:.1f} us, busy {busy / 1e3:.1f} us, gaps {(span - busy) / 1e3:.1f} us")


if len(sys.argv) > 2 and sys.argv[1] == "--report":
    report(sys.argv[2])
    sys.exit(0)

import torch
import ravqa_amd  # noqa: F401
from ravqa_amd import synth
from ravqa_amd.scorer import IndexScorer

bsz = int(sys.argv[1]) if len(sys.argv) > 1 else 1
corpus = synth.make_corpus(1_000_000, 128, 131072, 2, seed=0, device="cuda")
Q, _ = synth.make_queries(corpus, 64, 32, seed=2)
scorer = IndexScorer(device_index=synth.corpus_device_index(corpus), max_batch=256)
for i in range(60):
    scorer.search_batch(Q[i % 32:i % 32 + bsz], 100, 2, 0.45, 1024, 32)
    torch.cuda.synchronize()
scorer.check()
