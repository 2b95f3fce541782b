#!/bin/bash
# k_bfknn2_frames_mfma with eight wavefronts per workgroup (two train halves) against the four-wavefront form: tests, the launch
# by events (bench.py cross_camera), the cross-camera step
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_cross
mkdir -p $out
cd $root
timeout -k 10 600 python3 -m pytest tests/test_gpu_resident.py tests/test_gpu_multicam.py -m gpu -x -q 2>&1 | tail -2 || exit 1
run() { tag=$1; shift; env "$@" > $out/$tag.json 2> $out/$tag.err || { tail -3 $out/$tag.err; return; }; python3 -c "
import json; d = json.load(open('$out/$tag.json')); c = d['cross_camera']; print('$tag step', round(d['ms_per_step'],4), 'cross', round(c['ms_per_step'],4), 'knn', round(c['knn2_launch_ms'],4), 'frac', round(c['roofline']['frac'],3))"; }
B="python3 bench.py --no-cpu-baseline --no-pcie --no-pipelined"
run h1a ORBFE_KNN2_HALVES=1 $B
run h2a ORBFE_KNN2_HALVES=2 $B
run h1b ORBFE_KNN2_HALVES=1 $B
run h2b ORBFE_KNN2_HALVES=2 $B
