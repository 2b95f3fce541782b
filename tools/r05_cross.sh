#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_cross
mkdir -p $out
cd $root
timeout -k 10 600 python3 -m pytest tests/test_gpu_multicam.py -m gpu -x -q 2>&1 | tail -4
python3 bench.py --no-cpu-baseline --no-pcie --no-pipelined --steps 100 > $out/cross.json 2> $out/cross.err || { tail -5 $out/cross.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("$out/cross.json"))
print("step", d["ms_per_step"], json.dumps(d["cross_camera"])[:900])
PY
