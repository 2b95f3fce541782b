#!/usr/bin/env python3
"""Summarise rocprofv3 CSVs collected by tools/collect_profiles.sh.

Per kernel: mean counter value per dispatch (pmc_*/), mean duration (prof_trace/), and the HBM
traffic per launch = FETCH_SIZE*1024*corr + WRITE_SIZE*1024, where corr comes from the calibration
run (calib/): bytes actually read / (FETCH_SIZE*1024) for the access width the kernel uses.
"""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"


def kname(s):
    return s.split("(")[0].replace("void ", "")


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = kname(row["Kernel_Name"])
        if name.startswith("k_"):
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(root + "/prof_trace/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = kname(row["Kernel_Name"])
        if name.startswith("k_"):
            dur[name].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
calib = {}
for f in glob.glob(root + "/calib/**/*counter_collection.csv", recursive=True):
    tmp = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "FETCH_SIZE" and "k_read" in row["Kernel_Name"]:
            tmp[kname(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    for k, v in tmp.items():
        width = {"unsigned char": 1, "unsigned int": 4}.get(k.split("<")[1].rstrip(">"), 16)
        calib[width] = (256 << 20) / (1024.0 * v[-1])  # last (warm) repetition
# access width of each kernel's dominant global reads (bytes per lane)
WIDTH = {"k_fast_cells": 4, "k_orient_blur_desc": 4, "k_pyr_fused": 4, "k_pyr_level0": 1, "k_pyr_resize": 1,
         "k_octree": 4, "k_pack": 4}
out = {"workload": {"batch": 64, "rows": 480, "cols": 752, "nfeatures": 1000},
       "command": "python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline (one rocprofv3 --pmc pass per counter set)",
       "calibration_bytes_per_FETCH_SIZE_KB": calib, "kernels": {}}
for k, d in sorted(acc.items()):
    e = {c: sum(v) / len(v) for c, v in sorted(d.items())}
    e["dispatches"] = max(len(v) for v in d.values())
    base = k.split("<")[0]
    corr = calib.get(WIDTH.get(base, 4), 1.0)
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = e["FETCH_SIZE"] * 1024 * corr + e["WRITE_SIZE"] * 1024
        e["fetch_correction"] = corr
    if k in dur:
        e["avg_duration_us"] = sum(dur[k]) / len(dur[k]) / 1e3
        e["trace_calls"] = len(dur[k])
    out["kernels"][k] = e
print(json.dumps(out, indent=1))
