#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_dist
mkdir -p $out
cd $root
export ORBFE_BENCH_FORCE_DIST=1
run() { # tag, args...
  tag=$1; shift
  python3 bench.py --no-cpu-baseline --no-pcie --no-pipelined --no-cross --config c4 --batch 8 "$@" > $out/$tag.json 2> $out/$tag.err || { tail -5 $out/$tag.err; exit 1; }
  python3 - <<PY
import json
d = json.load(open("$out/$tag.json"))
print("$tag ms_per_step=%.4f" % d["ms_per_step"], flush=True)
PY
}
for pr in 1,1,1,1 0,0,0,0 0,1,0,1 0,1,-1,1; do
 for cp in -1 0 1; do
  for l in 2 3; do
   export ORBFE_LANE_PRIOS=$pr ORBFE_MC_COMM_PRIO=$cp
   run dist_p${pr//,/_}_c${cp}_l$l --lanes $l
  done
 done
done
export ORBFE_LANE_PRIOS=0,0,0,0 ORBFE_MC_COMM_PRIO=-1
run dist_q8_p0_l3 --lanes 3 --hw-queues 8
run dist_q8_p0_l4 --lanes 4 --hw-queues 8
export ORBFE_LANE_PRIOS=1,1,1,1
run dist_q8_p1_l3 --lanes 3 --hw-queues 8
