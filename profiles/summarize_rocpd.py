#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max duration) from a rocprofv3 `--kernel-trace --stats` rocpd database.
Usage: python profiles/summarize_rocpd.py gpurun_out/prof_rXX/bench_results.db > profiles/rXX_kernel_stats.csv"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                       "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(grid_x), max(grid_y), max(workgroup_x) "
                       "from kernels group by name order by 3 desc"))
total = sum(r[2] for r in rows)
print("kernel,calls,total_ms,pct_of_gpu_time,avg_us,min_us,max_us,vgpr,agpr,lds_bytes,grid_x,grid_y,block_x")
for r in rows:
    name = r[0].replace(",", ";")
    if len(sys.argv) > 2 and sys.argv[2] == "--ours" and "at::" in name or "rocprim" in name:
        continue
    print(f"\"{name[:120]}\",{r[1]},{r[2] / 1e6:.3f},{100.0 * r[2] / total:.2f},{r[3] / 1e3:.2f},{r[4] / 1e3:.2f},{r[5] / 1e3:.2f},"
          f"{r[6]},{r[7]},{r[8]},{r[9]},{r[10]},{r[11]}")
