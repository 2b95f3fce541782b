#!/bin/bash
# the whole GPU suite several times over in fresh processes, with tools/diag/abrt_bt.c preloaded; stops at the first failure
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_soak
mkdir -p $out
cd $root
gcc -shared -fPIC -O1 -o /tmp/libabrt_bt.so tools/diag/abrt_bt.c || exit 1
for k in 1 2 3 4 5 6; do
  LD_PRELOAD=/tmp/libabrt_bt.so timeout -k 10 600 python3 -m pytest -p no:faulthandler tests -m gpu -x -q > $out/full$k.log 2> $out/full$k.err
  rc=$?
  tail -1 $out/full$k.log
  echo "run $k rc=$rc" >> $out/progress6.log
  [ $rc = 0 ] || { grep -v "amdgpu.ids" $out/full$k.err | tail -60; tail -5 $out/full$k.log; exit $rc; }
done
