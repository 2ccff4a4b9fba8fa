#!/bin/bash
# SQ / TCC counters of the stage-2 kernels of ONE variant over a short bench run.  usage: bash profiles/pmc_s2_one.sh <impl> <outdir>
set -u
IMPL=$1; OUT=$(readlink -f "${2:-gpurun_out/pmc_s2_$1}"); mkdir -p "$OUT"
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
KREGEX='filter_stage2|s2_combine'
run() { name=$1; shift
  FLMR_S2_IMPL=$IMPL timeout 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$KREGEX" --output-format csv -d "$OUT/$name" -o p -- \
      python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/$name.log" 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA
run sq3 SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM SQ_WAVES
run tcc TCC_HIT_sum TCC_MISS_sum
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
cd $R
python profiles/summarize_pmc.py "$OUT" | tee "$OUT/summary.csv"
