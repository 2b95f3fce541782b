#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_pairs2
mkdir -p $out
cd $root
python3 - <<PY
import sys
sys.path.insert(0, "$root")
import bench
open("/tmp/frames.raw", "wb").write(bench.bench_frames(480, 752, 8).tobytes())
PY
run() { tag=$1; shift; env "$@" tools/hostbench /tmp/frames.raw 480 752 8 1200 0 stream > $out/$tag.json 2> $out/$tag.err; echo "$tag: $(cat $out/$tag.json | cut -c1-600)"; }
run base A=1
run nospin ORBFE_SPIN=0
run prio0 ORBFE_LANE_PRIOS=0,0,0,0
run prio1 ORBFE_LANE_PRIOS=1,1,1,1
run nozc ORBFE_ZEROCOPY=0
run noupk ORBFE_UPLOAD_KERNEL=0
run nospin_prio0 ORBFE_SPIN=0 ORBFE_LANE_PRIOS=0,0,0,0
run q8 GPU_MAX_HW_QUEUES=8
