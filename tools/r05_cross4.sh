#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_cross
mkdir -p $out
rm -rf $out/trace3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/trace3 -- python3 $root/bench.py --no-cpu-baseline --no-pcie --no-pipelined --steps 200 > $out/trace3.json 2> $out/trace3.err
python3 - <<PY
import json, csv, glob
d = json.load(open("$out/trace3.json")); print("step", d["ms_per_step"], "cross", d["cross_camera"]["ms_per_step"])
rows = []
for f in glob.glob("$out/trace3/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
cp = []
for f in glob.glob("$out/trace3/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        cp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Size", r.get("Bytes", "")), "-", "-"))
rows += cp
rows.sort()
knn = [i for i, r in enumerate(rows) if "bfknn2_frames" in r[2]]
print("knn launches", len(knn), "copies", len(cp))
# the cross region: the knn launches that are followed by k_pyr (not the 20 back-to-back timing launches)
sel = knn[60:70]
t0 = rows[sel[0]][0]
for i in range(sel[0], sel[-1] + 1):
    s, e, n, q, st = rows[i]
    print("%9.1f %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
PY
