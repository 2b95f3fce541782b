#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_cross
mkdir -p $out
cd $root
run() { tag=$1; shift; env "$@" > $out/$tag.json 2> $out/$tag.err || { tail -3 $out/$tag.err; return; }; python3 -c "
import json; d = json.load(open('$out/$tag.json')); print('$tag step', round(d['ms_per_step'],4), 'cross', round(d['cross_camera']['ms_per_step'],4))"; }
B="python3 bench.py --no-cpu-baseline --no-pcie --no-pipelined"
run s100 A=1 $B --steps 100
run s200 A=1 $B --steps 200
run s600 A=1 $B --steps 600
run s300_vec ORBFE_KNN2_MFMA=0 $B --steps 300
run s300_l1 A=1 $B --steps 300 --lanes 1
run s300_l3 A=1 $B --steps 300 --lanes 3
run s300_rot1 A=1 $B --steps 300 --rotate 1
run s300_q8 A=1 $B --steps 300 --hw-queues 8
