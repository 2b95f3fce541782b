#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_knn
mkdir -p $out
cd $root
for kb in 0 82 88 96 112 128 144 160; do
ORBFE_KNN2_LDS_KB=$kb python3 bench.py --no-cpu-baseline --no-pcie --no-pipelined --steps 60 > $out/cross_lds$kb.json 2> $out/cross_lds$kb.err || { tail -5 $out/cross_lds$kb.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("$out/cross_lds$kb.json"))
print("lds_kb=$kb knn2_launch_ms=%.5f cross ms_per_step %.4f" % (d["cross_camera"]["knn2_launch_ms"], d["cross_camera"]["ms_per_step"]))
PY
done
