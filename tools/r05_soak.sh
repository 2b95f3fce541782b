#!/bin/bash
# soak: the tests of the round's host-side machinery (lanes, frames in flight, exchange, handles under threads) several times over
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_soak
mkdir -p $out
cd $root
for k in 1 2 3 4 5 6; do
  timeout -k 10 400 python3 -m pytest tests/test_gpu_lanes.py tests/test_gpu_hostpath.py tests/test_gpu_multicam.py tests/test_gpu_keyframes.py -m gpu -x -q > $out/run$k.log 2>&1
  rc=$?
  tail -1 $out/run$k.log
  [ $rc = 0 ] || { tail -30 $out/run$k.log; exit $rc; }
done
