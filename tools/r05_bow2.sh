#!/bin/bash
# where the host time of SearchByBoW x 64 goes (a -DORBFE_CALL_TRACE build under build_ab/trace, ORBFE_CALL_TRACE=1)
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_bow
mkdir -p $out
cd $root
python3 - <<PY
import sys
sys.path.insert(0, "$root")
import bench
open("/tmp/frames.raw", "wb").write(bench.bench_frames(480, 752, 8).tobytes())
PY
for mode in 0 1; do
  LD_LIBRARY_PATH=$root/build_ab/trace:$LD_LIBRARY_PATH ORBFE_CALL_TRACE=1 ORBFE_BOW_DEVNODES=$mode timeout -k 10 300 tools/hostbench /tmp/frames.raw 480 752 8 1200 0 matcher > $out/trace_dev$mode.json 2> $out/trace_dev$mode.err
  echo "devnodes $mode"; grep "bow_run count=64" $out/trace_dev$mode.err | tail -124 | awk '{for(i=1;i<=NF;i++){if($i=="pass1")a+=$(i+1);if($i=="stage")b+=$(i+1);if($i=="launch")c+=$(i+1);if($i=="sync")d+=$(i+1);if($i=="tail")e+=$(i+1)};n++} END{printf "n=%d pass1 %.1f stage %.1f launch %.1f sync %.1f tail %.1f\n",n,a/n,b/n,c/n,d/n,e/n}'
done
