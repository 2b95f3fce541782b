#!/bin/bash
# the whole GPU suite N times, one process after the other, stopping at the first run that is not green (stderr is not captured:
# pytest.ini runs with --capture=sys, so a fatal message of the runtime would be in the log)
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r06_suite_soak
mkdir -p $out
cd $root
n=${1:-6}
for i in $(seq 1 $n); do
  timeout -k 10 400 python3 -m pytest tests -m gpu -x -q > $out/run$i.log 2>&1; rc=$?
  echo "run $i: rc $rc: $(tail -1 $out/run$i.log)" | tee -a $out/summary.txt
  [ $rc = 0 ] || exit $rc
done
