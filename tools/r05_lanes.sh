#!/bin/bash
# round 5: batch lanes -- parity tests of the new paths, then the lane sweeps (C4's 8-frame shard and the 64-frame headline)
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_lanes
mkdir -p $out
cd $root
timeout -k 10 900 python3 -m pytest tests/test_gpu_lanes.py tests/test_gpu_natural.py tests/test_golden.py -m gpu -x -q > $out/pytest.log 2>&1
rc=$?
tail -5 $out/pytest.log
[ $rc = 0 ] || exit $rc
run() { # tag, args...
  tag=$1; shift
  python3 bench.py --no-cpu-baseline --no-pcie --no-cross --no-pipelined "$@" > $out/$tag.json 2> $out/$tag.err || { tail -5 $out/$tag.err; exit 1; }
  python3 - <<PY
import json
d = json.load(open("$out/$tag.json"))
sb = d.get("same_batch") or {}
print("$tag ms_per_step=%.4f same_batch=%.4f value=%.1fM" % (d["ms_per_step"], sb.get("ms_per_step", 0), d["value"] / 1e6), flush=True)
PY
}
for l in 1 2 3 4; do run c4b8_l$l --config c4 --batch 8 --lanes $l; done
run c4b8_l2split --config c4 --batch 8 --lanes 2 --lane-mode split
for l in 2 3 4; do run c4b8_q8_l$l --config c4 --batch 8 --lanes $l --hw-queues 8; done
ORBFE_LANES_INPUT_GUARD=0 run c4b8_l3_noguard --config c4 --batch 8 --lanes 3
ORBFE_LANES_INPUT_GUARD=0 run c4b8_q8_l4_noguard --config c4 --batch 8 --lanes 4 --hw-queues 8
for l in 1 2 3 4; do run c2_l$l --lanes $l; done
run c2_l2split --lanes 2 --lane-mode split
run c2_l2split_same --lanes 2 --lane-mode split --rotate 1
for l in 2 3; do run c2_drv_l$l --lanes $l --steps 20 --warmup 5; done
run c2_drv_l2split --lanes 2 --lane-mode split --steps 20 --warmup 5
run c4_l2 --config c4 --lanes 2
run c4_l3 --config c4 --lanes 3
