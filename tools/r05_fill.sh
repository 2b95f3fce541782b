#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_cross
mkdir -p $out
cd $root
timeout -k 10 600 python3 -m pytest tests/test_gpu_multicam.py tests/test_gpu_resident.py -m gpu -x -q 2>&1 | tail -2
ORBFE_KNN2_MFMA=0 timeout -k 10 600 python3 -m pytest tests/test_gpu_multicam.py -m gpu -x -q 2>&1 | tail -2
run() { tag=$1; shift; env "$@" > $out/$tag.json 2> $out/$tag.err || { tail -3 $out/$tag.err; return; }; python3 -c "
import json; d = json.load(open('$out/$tag.json')); c = d['cross_camera']; print('$tag step', round(d['ms_per_step'],4), 'cross', round(c['ms_per_step'],4), 'knn', round(c['knn2_launch_ms'],4), 'frac', round(c['roofline']['frac'],3))"; }
B="python3 bench.py --no-cpu-baseline --no-pcie --no-pipelined"
run fill1 A=1 $B
run fill2 A=1 $B
