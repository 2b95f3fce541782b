#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_dist2
mkdir -p $out
cd $root
export ORBFE_BENCH_FORCE_DIST=1
run() { # tag, args...
  tag=$1; shift
  python3 bench.py --no-cpu-baseline --no-pcie --no-pipelined --no-cross "$@" > $out/$tag.json 2> $out/$tag.err || { tail -5 $out/$tag.err; exit 1; }
  python3 - <<PY
import json
d = json.load(open("$out/$tag.json"))
print("$tag ms_per_step=%.4f same=%.4f" % (d["ms_per_step"], (d.get("same_batch") or {}).get("ms_per_step", 0)), flush=True)
PY
}
for pr in 1,1,1,1 0,1,0,1 0,0,0,0; do
  export ORBFE_LANE_PRIOS=$pr
  run c2dist_p${pr//,/_}_l2 --lanes 2
  run c2dist_p${pr//,/_}_l2_again --lanes 2
  run c2dist_p${pr//,/_}_l3 --lanes 3
done
unset ORBFE_LANE_PRIOS
run c2dist_split --lanes 2 --lane-mode split
run c2dist_l1 --lanes 1
export ORBFE_LANE_PRIOS=0,1,0,1
run c4b8dist_l2_a --config c4 --batch 8 --lanes 2
run c4b8dist_l2_b --config c4 --batch 8 --lanes 2
run c4b8dist_l2_c --config c4 --batch 8 --lanes 2
