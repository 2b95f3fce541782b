#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_lanes3
mkdir -p $out
cd $root
run() { # tag, args...
  tag=$1; shift
  python3 bench.py --no-cpu-baseline --no-pcie --no-pipelined "$@" > $out/$tag.json 2> $out/$tag.err || { tail -5 $out/$tag.err; exit 1; }
  python3 - <<PY
import json
d = json.load(open("$out/$tag.json"))
sb = d.get("same_batch") or {}
print("$tag ms_per_step=%.4f same_batch=%.4f value=%.1fM cross=%s" % (d["ms_per_step"], sb.get("ms_per_step", 0), d["value"] / 1e6, (d.get("cross_camera") or {}).get("ms_per_step")), flush=True)
PY
}
for l in 2 3 4; do run c4b8_l${l} --config c4 --batch 8 --lanes $l --no-cross; done
for l in 3 4; do run c4b8_l${l}_g1 --config c4 --batch 8 --lanes $l --input-guard 1 --no-cross; done
export ORBFE_BENCH_FORCE_DIST=1
for l in 1 2 3 4; do run c4b8_dist_l${l} --config c4 --batch 8 --lanes $l; done
unset ORBFE_BENCH_FORCE_DIST
run c2_default --no-cross
run c2_drv --steps 20 --warmup 5
python3 - <<PY
import sys
sys.path.insert(0, "$root")
import bench
open("/tmp/frames.raw", "wb").write(bench.bench_frames(480, 752, 8).tobytes())
PY
tools/hostbench /tmp/frames.raw 480 752 8 1200 0 stream > $out/stream.json 2> $out/stream.err; cat $out/stream.json
