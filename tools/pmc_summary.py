#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (gpurun_out/pmc_*/**/*counter_collection.csv) per kernel:
mean counter value per dispatch."""
import csv
import glob
import collections
import json
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0]
        if not name.startswith("k_") and "k_orient" not in name:
            continue
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, d in sorted(acc.items()):
    out[k] = {c: sum(v) / len(v) for c, v in sorted(d.items())}
    out[k]["dispatches"] = max(len(v) for v in d.values())
print(json.dumps(out, indent=1))
