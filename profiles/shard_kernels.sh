# per-kernel times of rank 0's share of one exact-protocol step at N shards (default 8), 1024 queries: where the fixed costs are
N=${1:-8}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sm -o s -- python $R/profiles/shard_step_model.py 1000000 $N 1024 > /tmp/sm.log 2>&1
tail -3 /tmp/sm.log
python3 - <<'PY'
import csv,glob
f=glob.glob("/tmp/sm/**/s_kernel_stats.csv",recursive=True)[0]
rows=[r for r in csv.reader(open(f))][1:]
rows=[r for r in rows if not r[0].startswith("void at::") and "rocprim" not in r[0] and "rocclr" not in r[0] and "elementwise" not in r[0]]
rows.sort(key=lambda r:-float(r[2]))
for r in rows[:24]: print(r[0][:58].ljust(58), r[1].rjust(5), ("%.3f"%(float(r[3])/1e6)).rjust(8), "ms avg")
PY
