#!/bin/bash
# SQ counters of one kernel (regex) over a short bench run: where do its wave cycles go (issue / wait / LDS conflicts)?
# usage (GPU box, repo root):  bash profiles/pmc_one_kernel.sh '<kernel regex>' gpurun_out/pmc_x
set -u
KREGEX=${1:?kernel regex}
OUT=$(readlink -f "${2:-gpurun_out/pmc_one}"); mkdir -p "$OUT"
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$KREGEX" --output-format csv -d "$OUT/$name" -o p -- \
      python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras ${BENCH_ARGS:-} > "$OUT/$name.log" 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU
cd $R
python profiles/summarize_pmc.py "$OUT"
