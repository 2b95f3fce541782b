#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_pairs3
mkdir -p $out
cd $root
timeout -k 10 900 python3 -m pytest tests/test_gpu_hostpath.py tests/test_gpu_lanes.py -m gpu -x -q > $out/pytest.log 2>&1
rc=$?
tail -3 $out/pytest.log
[ $rc = 0 ] || exit $rc
python3 - <<PY
import sys
sys.path.insert(0, "$root")
import bench
open("/tmp/frames.raw", "wb").write(bench.bench_frames(480, 752, 8).tobytes())
PY
run() { tag=$1; shift; env "$@" tools/hostbench /tmp/frames.raw 480 752 8 1200 0 stream > $out/$tag.json 2> $out/$tag.err; echo "$tag: $(cat $out/$tag.json | cut -c1-600)"; }
run base A=1
run nospin ORBFE_SPIN=0
run q8 GPU_MAX_HW_QUEUES=8
python3 bench.py --config c3 --no-cpu-baseline > $out/c3.json 2> $out/c3.err || { tail -5 $out/c3.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("$out/c3.json"))
print(json.dumps(d["protocols_ms"])[:1500])
PY
