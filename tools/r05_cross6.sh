#!/bin/bash
# the cross-camera leg, 300 steps, with the interpreter's collector off (bench.py): knn-2 launch padded to 96 KB LDS or not
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_cross
mkdir -p $out
cd $root
run() { tag=$1; shift; env "$@" > $out/$tag.json 2> $out/$tag.err || { tail -3 $out/$tag.err; return; }; python3 -c "
import json; d = json.load(open('$out/$tag.json')); print('$tag step', round(d['ms_per_step'],4), 'cross', round(d['cross_camera']['ms_per_step'],4), 'knn', round(d['cross_camera']['knn2_launch_ms'],4))"; grep "cross worst" $out/$tag.err | cut -c1-400; }
B="python3 bench.py --no-cpu-baseline --no-pcie --no-pipelined"
run pad1 ORBFE_BENCH_CROSS_TRACE=1 $B --steps 300
run nopad1 ORBFE_BENCH_CROSS_TRACE=1 ORBFE_KNN2_PAD=0 $B --steps 300
run pad2 A=1 $B --steps 300
run nopad2 ORBFE_KNN2_PAD=0 $B --steps 300
run pad3 A=1 $B --steps 1000
run nopad3 ORBFE_KNN2_PAD=0 $B --steps 1000
