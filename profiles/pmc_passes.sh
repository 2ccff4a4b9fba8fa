#!/bin/bash
# rocprofv3 PMC passes over bench.py (counters in their own runs; one TCC-heavy counter per pass as the guide prescribes).
# usage (on the GPU box, from the repo root):  bash profiles/pmc_passes.sh gpurun_out/pmc_rXX
set -u
OUT=$(readlink -f "$1"); mkdir -p "$OUT"
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
KREGEX='s0_|filter_stage|s2_combine|maxsim|select_topn|sort_topn|s1_|cand_|qualifying'
run() { # name, counters...
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$KREGEX" --output-format csv -d "$OUT/$name" -o p -- \
      python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/$name.log" 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum
cd $R
python profiles/summarize_pmc.py "$OUT" > "$OUT/summary.csv"
