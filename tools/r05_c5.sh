#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_c5
mkdir -p $out
cd $root
python3 tools/c5_stages.py 2>&1 | tail -3
python3 - <<PY
import sys
sys.path.insert(0, "$root")
import bench
open("/tmp/frames1024.raw", "wb").write(bench.bench_frames(1024, 1024, 8).tobytes())
PY
tools/hostbench /tmp/frames1024.raw 1024 1024 8 1500 0 c5 | cut -c1-900
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $root/tools/hostbench /tmp/frames1024.raw 1024 1024 8 1500 0 c5 > $out/trace.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$out/trace/*/*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        print("%-70s calls %5s avg %8.1f us total %6.1f%%" % (row["Name"][:70], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["Percentage"])))
PY
