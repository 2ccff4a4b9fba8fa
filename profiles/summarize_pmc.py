#!/usr/bin/env python3
"""Collapse the rocprofv3 counter_collection CSVs of profiles/pmc_passes.sh into one row per kernel:
mean counter value per dispatch (summed over the dimension instances rocprofv3 emits)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))   # kernel -> counter -> sum over dispatches
ndisp = defaultdict(lambda: defaultdict(set))
for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0][:60]
            c = row["Counter_Name"]
            acc[k][c] += float(row["Counter_Value"])
            ndisp[k][c].add(row["Dispatch_Id"])
counters = sorted({c for k in acc for c in acc[k]})
print("kernel," + ",".join(counters))
for k in sorted(acc):
    print(k.replace(",", ";") + "," + ",".join(f"{acc[k][c] / max(1, len(ndisp[k][c])):.4g}" if c in acc[k] else "" for c in counters))
