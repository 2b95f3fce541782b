#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_cross
mkdir -p $out
cd $root
run() { tag=$1; shift; env "$@" > $out/$tag.json 2> $out/$tag.err || { tail -3 $out/$tag.err; return; }; python3 -c "
import json; d = json.load(open('$out/$tag.json')); print('$tag step', round(d['ms_per_step'],4), 'cross', round(d['cross_camera']['ms_per_step'],4), d['cross_camera'].get('knn2_launch_ms'))"; }
B="python3 bench.py --no-cpu-baseline --no-pcie --no-pipelined"
run pad_s100 A=1 $B --steps 100
run pad_s300 A=1 $B --steps 300
run pad_s300b A=1 $B --steps 300
run nopad_s100 ORBFE_KNN2_PAD=0 $B --steps 100
run nopad_s300 ORBFE_KNN2_PAD=0 $B --steps 300
run nopad_s300b ORBFE_KNN2_PAD=0 $B --steps 300
run vec_s300 ORBFE_KNN2_MFMA=0 $B --steps 300
