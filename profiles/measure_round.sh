#!/bin/bash
# One round's evidence in one call (GPU box, repo root): bench JSON, rocprofv3 kernel stats, PMC passes.
# usage: bash profiles/measure_round.sh gpurun_out/rXX
set -u
mkdir -p "$1"
OUT=$(readlink -f "$1")
R=$(pwd)
timeout 1500 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 600 "$OUT/bench.json"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$OUT/stats.log" 2>&1
cd $R
bash profiles/pmc_passes.sh "$OUT/pmc" > "$OUT/pmc.log" 2>&1
ls "$OUT" "$OUT/stats" | head -30
