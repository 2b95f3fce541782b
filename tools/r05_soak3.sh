#!/bin/bash
# as tools/r05_soak.sh (the four files in one process), with the runtime's error log (AMD_LOG_LEVEL=1), glibc's fatal messages and
# RCCL's warnings captured; stops at the first failure
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_soak
mkdir -p $out
cd $root
export LIBC_FATAL_STDERR_=1 AMD_LOG_LEVEL=1 NCCL_DEBUG=WARN PYTHONFAULTHANDLER=1
for k in 1 2 3 4 5 6; do
  timeout -k 10 400 python3 -X faulthandler -m pytest tests/test_gpu_lanes.py tests/test_gpu_hostpath.py tests/test_gpu_multicam.py tests/test_gpu_keyframes.py -m gpu -x -q -v > $out/all$k.log 2> $out/all$k.err
  rc=$?
  tail -1 $out/all$k.log
  echo "run $k rc=$rc" >> $out/progress3.log
  [ $rc = 0 ] || { grep -v "^  File" $out/all$k.err | tail -40; tail -5 $out/all$k.log; exit $rc; }
done
