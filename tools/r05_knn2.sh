#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_knn
mkdir -p $out
cd $root
for kb in 0 96 128; do
ORBFE_KNN2_LDS_KB=$kb python3 bench.py --no-cpu-baseline --no-pcie --no-pipelined --steps 100 > $out/cross_lds$kb.json 2> $out/cross_lds$kb.err || { tail -5 $out/cross_lds$kb.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("$out/cross_lds$kb.json"))
print("lds_kb=$kb knn2_launch_ms=%.5f" % d["cross_camera"]["knn2_launch_ms"])
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/tools/knn_times.py > $out/trace.log 2>&1
grep -h "bfknn2" $out/trace/*/*kernel_stats.csv
