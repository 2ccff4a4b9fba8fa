#!/bin/bash
# rocprofv3 PMC passes over the stand-alone dense stage-1 probe (regime by the S1D_* environment).
# usage (GPU box, repo root):  S1D_N=8700 ... bash profiles/pmc_s1_probe.sh gpurun_out/r06/pmc_probe
set -u
mkdir -p "$1"; OUT=$(readlink -f "$1")
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex 's1_' --output-format csv -d "$OUT/$name" -o p -- \
      $R/profiles/microbench/s1_dense_probe > "$OUT/$name.log" 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
run sq2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
cd $R
python - "$OUT" <<'PY'
import sys,glob,csv,collections
out=sys.argv[1]
for d in ['sq1','sq2','tcc','fetch']:
    for f in glob.glob(f'{out}/{d}/**/*counter_collection.csv',recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:40]; acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        for k,v in acc.items():
            print(d,k,{c:f'{x:.4g}' for c,x in v.items()})
PY
