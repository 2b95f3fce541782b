#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_cross
mkdir -p $out
cd $root
for st in 100 300 300; do
python3 bench.py --no-cpu-baseline --no-pcie --no-pipelined --steps $st > $out/cross_s$st.json 2> $out/cross_s$st.err || { tail -5 $out/cross_s$st.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("$out/cross_s$st.json"))
print("steps $st step", d["ms_per_step"], "cross", d["cross_camera"]["ms_per_step"], d["cross_camera"].get("knn2_launch_ms"))
PY
done
