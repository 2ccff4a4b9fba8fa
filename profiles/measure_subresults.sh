#!/bin/bash
# rocprofv3 kernel statistics of every sub-result workload of bench.py + one PMC pass of the two dense stage-1 workloads: what the
# numbers of README's table are reproducible from.  usage (GPU box, repo root): bash profiles/measure_subresults.sh gpurun_out/r06/sub
set -u
OUT=$(readlink -f "$1"); mkdir -p "$OUT"
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for w in ${SUBS:-k500 thr0.25 nq320 nq832 built4096 built256}; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$w" -o s -- python $R/profiles/sub_result_probe.py $w > "$OUT/$w.log" 2>&1
  grep '^{' "$OUT/$w.log" | tail -1 > "$OUT/$w.json"
  f=$(ls "$OUT/$w"/s_kernel_stats.csv "$OUT/$w"/*/s_kernel_stats.csv 2>/dev/null | head -1)
  # the path's own kernels only (the corpus generator's torch kernels are not part of the step), top 14 by total time
  if [ -n "$f" ]; then (head -1 "$f"; grep -E 's0_|filter_stage|s2_|maxsim|s3_|select_topn|sort_topn|s1_|cand_|qualifying' "$f" | head -14) > "$OUT/${w}_kernel_stats.csv"; fi
done
KREGEX='s1_image|s1_exact|select_topn|filter_stage2_xcd|maxsim'
for w in ${PMCS-thr0.25 built256}; do
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-include-regex "$KREGEX" --output-format csv -d "$OUT/pmc_$w/sq" -o p -- python $R/profiles/sub_result_probe.py $w > "$OUT/pmc_$w.sq.log" 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-include-regex "$KREGEX" --output-format csv -d "$OUT/pmc_$w/fetch" -o p -- python $R/profiles/sub_result_probe.py $w > "$OUT/pmc_$w.fetch.log" 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "$KREGEX" --output-format csv -d "$OUT/pmc_$w/tcc" -o p -- python $R/profiles/sub_result_probe.py $w > "$OUT/pmc_$w.tcc.log" 2>&1
  python3 $R/profiles/summarize_pmc.py "$OUT/pmc_$w" > "$OUT/pmc_${w}_summary.csv"
done
cd $R
ls "$OUT"
