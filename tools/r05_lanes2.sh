#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_lanes2
mkdir -p $out
cd $root
run() { # tag, args...
  tag=$1; shift
  python3 bench.py --no-cpu-baseline --no-pcie --no-cross --no-pipelined "$@" > $out/$tag.json 2> $out/$tag.err || { tail -5 $out/$tag.err; exit 1; }
  python3 - <<PY
import json
d = json.load(open("$out/$tag.json"))
sb = d.get("same_batch") or {}
sf = d.get("single_frame") or {}
print("$tag ms_per_step=%.4f same_batch=%.4f value=%.1fM" % (d["ms_per_step"], sb.get("ms_per_step", 0), d["value"] / 1e6), flush=True)
PY
}
for l in 2 3 4; do run c4b8_l${l}_g0 --config c4 --batch 8 --lanes $l; done
for l in 2 3 4; do run c4b8_l${l}_g1 --config c4 --batch 8 --lanes $l --input-guard 1; done
for l in 3 4; do run c4b8_q8_l${l}_g0 --config c4 --batch 8 --lanes $l --hw-queues 8; done
export ORBFE_LANE_PRIOS=0,0,0,0
for l in 3 4; do run c4b8_p0_l${l}_g0 --config c4 --batch 8 --lanes $l; done
for l in 3 4; do run c4b8_p0_q8_l${l}_g0 --config c4 --batch 8 --lanes $l --hw-queues 8; done
export ORBFE_LANE_PRIOS=1,1,1,1
for l in 3 4; do run c4b8_p1_l${l}_g0 --config c4 --batch 8 --lanes $l; done
unset ORBFE_LANE_PRIOS
ORBFE_LANES_FORK_ALWAYS=1 run c4b8_l3_g0_fork --config c4 --batch 8 --lanes 3
for l in 2 3 4; do run c2_l${l}_g0 --lanes $l; done
for l in 2 3; do run c2_drv_l${l}_g0 --lanes $l --steps 20 --warmup 5; done
run c2_drv_l2split --lanes 2 --lane-mode split --steps 20 --warmup 5
run c2_drv_l2split_r1 --lanes 2 --lane-mode split --steps 20 --warmup 5 --rotate 1
