#!/bin/bash
# randomised extractor sweep (tools/stress_parity.py: sizes, pyramids, thresholds, lapping ranges, batches, content kinds incl.
# photographs) on the final tree of round 5
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_stress
mkdir -p $out
cd $root
for seed in 611 612; do
  timeout -k 10 520 python3 tools/stress_parity.py 150 $seed > $out/parity_$seed.log 2>&1 || { tail -5 $out/parity_$seed.log; exit 1; }
  tail -1 $out/parity_$seed.log
done
