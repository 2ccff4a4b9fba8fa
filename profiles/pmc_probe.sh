#!/bin/bash
# SQ counters of one kernel (regex) over the stand-alone S3 harness.  usage (GPU box, repo root):
#   S3P_NQ=832 bash profiles/pmc_probe.sh 'maxsim_qs' gpurun_out/pmc_qs profiles/microbench/s3_probe qs
set -u
KREGEX=${1:?kernel regex}; OUT=$(readlink -f "$2"); shift 2
mkdir -p "$OUT"; R=$(pwd); CMD="$R/$1"; shift
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$KREGEX" --output-format csv -d "$OUT/$name" -o p -- $CMD $ARGS > "$OUT/$name.log" 2>&1
}
ARGS="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for name in ("sq1", "sq2"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            acc[(row["Kernel_Name"][:40], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(f"{k:42s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
