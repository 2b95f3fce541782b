#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_knn
mkdir -p $out
cd $root
timeout -k 10 600 python3 -m pytest tests/test_gpu_resident.py tests/test_gpu_multicam.py -m gpu -x -q 2>&1 | tail -3
python3 bench.py --no-cpu-baseline --no-pcie --no-pipelined --steps 100 > $out/cross_final.json 2> $out/cross_final.err || { tail -5 $out/cross_final.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("$out/cross_final.json"))
c = d["cross_camera"]
print("knn2_launch_ms=%.5f cross ms_per_step %.4f frac %.3f" % (c["knn2_launch_ms"], c["ms_per_step"], c["roofline"]["frac"]))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace2 -- python3 $root/bench.py --no-cpu-baseline --no-pcie --no-pipelined --steps 60 > $out/trace2.log 2>&1
grep -h "bfknn2" $out/trace2/*/*kernel_stats.csv
