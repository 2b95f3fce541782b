#!/bin/bash
# C5 after the matching call's inputs are read in place and its results stored into the pinned mirror: tests, hostbench c5, timeline
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_c5
mkdir -p $out
rm -rf $out/trace2
cd $root
timeout -k 10 600 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_matcher.py -m gpu -x -q > $out/pytest.log 2>&1
rc=$?
tail -3 $out/pytest.log
[ $rc = 0 ] || exit $rc
python3 - <<PY
import sys
sys.path.insert(0, "$root")
import bench
open("/tmp/frames1024.raw", "wb").write(bench.bench_frames(1024, 1024, 8).tobytes())
PY
for k in 1 2; do tools/hostbench /tmp/frames1024.raw 1024 1024 8 1500 0 c5 | cut -c1-1200; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/trace2 -- $root/tools/hostbench /tmp/frames1024.raw 1024 1024 8 1500 0 c5 > $out/trace2.log 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$out/trace2/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50].replace("(anonymous namespace)::", "")))
for f in glob.glob("$out/trace2/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
rows.sort()
idx = [i for i, r in enumerate(rows) if "k_fisheye_stereo" in r[2]]
i0 = idx[-3]
t0 = rows[i0 - 12][0]
for r in rows[i0 - 12: i0 + 3]:
    print("%9.1f %8.1f  %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[2]))
PY
