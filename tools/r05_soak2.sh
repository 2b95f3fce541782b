#!/bin/bash
# diagnostic for the abort of tools/r05_soak.sh (run 3: SIGABRT without a message while the main thread was in the oracle, in the
# fourth test of tests/test_gpu_multicam.py): the same tests with glibc's and the runtime's fatal messages captured
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_soak
mkdir -p $out
cd $root
export LIBC_FATAL_STDERR_=1 AMD_LOG_LEVEL=1 NCCL_DEBUG=WARN PYTHONFAULTHANDLER=1
for k in 1 2 3 4 5 6 7 8; do
  timeout -k 10 300 python3 -X faulthandler -m pytest tests/test_gpu_multicam.py -m gpu -x -q > $out/mc$k.log 2> $out/mc$k.err
  rc=$?
  tail -1 $out/mc$k.log
  echo "run $k rc=$rc" >> $out/progress2.log
  [ $rc = 0 ] || { tail -40 $out/mc$k.err; exit $rc; }
done
