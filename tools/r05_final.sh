#!/bin/bash
# round 5: the numbers that go into profiles/r05_* (run on the GPU box through gpurun; copy the summaries afterwards)
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_final
mkdir -p $out
cd $root
if [ "$1" != "notests" ]; then
  timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1
  rc=$?
  tail -3 $out/pytest.log
  [ $rc = 0 ] || exit $rc
fi
b() { tag=$1; shift; python3 bench.py "$@" > $out/$tag.json 2> $out/$tag.err || { tail -5 $out/$tag.err; exit 1; }; python3 -c "
import json; d = json.load(open('$out/$tag.json')); print('$tag', d['ms_per_step'], '%.1f M/s' % (d['value'] / 1e6), (d.get('same_batch') or {}).get('ms_per_step'))"; }
b bench_752x480_b64
b bench_752x480_b64_steps20_warmup5 --steps 20 --warmup 5
b bench_c3_stereo_pair --config c3
b bench_c4_1280x720_b64 --config c4 --no-cpu-baseline
b bench_c4_1280x720_b8 --config c4 --batch 8 --no-cpu-baseline
b bench_c5_fisheye_pair --config c5
ORBFE_BENCH_FORCE_DIST=1 python3 bench.py --config c4 --batch 8 --no-cpu-baseline --no-pcie --no-pipelined > $out/bench_c4_1280x720_b8_rccl_world1.json 2> $out/rccl.err || tail -3 $out/rccl.err
python3 - <<PY
import sys
sys.path.insert(0, "$root")
import bench
open("/tmp/frames.raw", "wb").write(bench.bench_frames(480, 752, 8).tobytes())
PY
tools/hostbench /tmp/frames.raw 480 752 8 1200 0 matcher > $out/matcher_hostbench.json 2> $out/matcher_hostbench.err || tail -3 $out/matcher_hostbench.err
tools/hostbench /tmp/frames.raw 480 752 8 1200 0 stream > $out/stream_hostbench.json 2> $out/stream_hostbench.err || tail -3 $out/stream_hostbench.err
echo benches done
bash tools/collect_profiles.sh > $out/collect.log 2>&1
echo profiles done
