#!/bin/bash
# kernel / copy timeline of SearchByBoW x 64 (hostbench matcher under rocprofv3 --kernel-trace --memory-copy-trace)
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_bow
mkdir -p $out
rm -rf $out/trace
cd $root
python3 - <<PY
import sys
sys.path.insert(0, "$root")
import bench
open("/tmp/frames.raw", "wb").write(bench.bench_frames(480, 752, 8).tobytes())
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/trace -- $root/tools/hostbench /tmp/frames.raw 480 752 8 1200 0 matcher > $out/trace.json 2> $out/trace.err
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$out/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Grid_Size_Y", 1) or 1)))
for f in glob.glob("$out/trace/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""), 0, 0))
rows.sort()
big = [i for i, r in enumerate(rows) if "k_search_bow" in r[2] and (r[4] == 64 or r[3] > 100 * 256)]
print("device-paired launches", len(big))
i0 = big[len(big) // 2]
t0 = rows[i0 - 3][0]
for r in rows[i0 - 3: i0 + 9]:
    print("%9.1f %8.1f  %s grid %d x %d" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[2], r[3], r[4]))
PY
