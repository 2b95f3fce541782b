#!/bin/bash
# the four files of tools/r05_soak.sh again, with a preloaded handler that prints the C backtrace of whoever raises SIGABRT
# (tools/diag/abrt_bt.c; pytest's own faulthandler off, it would replace the handler); stops at the first failure
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_soak
mkdir -p $out
cd $root
gcc -shared -fPIC -O1 -o /tmp/libabrt_bt.so tools/diag/abrt_bt.c || exit 1
export LIBC_FATAL_STDERR_=1
for k in 1 2 3 4 5 6 7 8 9 10 11 12; do
  LD_PRELOAD=/tmp/libabrt_bt.so timeout -k 10 400 python3 -m pytest -p no:faulthandler tests/test_gpu_lanes.py tests/test_gpu_hostpath.py tests/test_gpu_multicam.py tests/test_gpu_keyframes.py -m gpu -x -q > $out/bt$k.log 2> $out/bt$k.err
  rc=$?
  tail -1 $out/bt$k.log
  echo "run $k rc=$rc" >> $out/progress5.log
  [ $rc = 0 ] || { grep -v "amdgpu.ids" $out/bt$k.err | tail -60; exit $rc; }
done
