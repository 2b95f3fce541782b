#!/bin/bash
# randomised matcher sweep on the final tree of round 5 (kind 10 now mixes keyframe handles into its large SearchByBoW batches)
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_stress
mkdir -p $out
cd $root
for seed in 501 502 503; do
  STRESS_QUIET=1 timeout -k 10 500 python3 tools/stress_matcher.py 1100 $seed > $out/matcher_$seed.log 2>&1 || { tail -5 $out/matcher_$seed.log; exit 1; }
  tail -1 $out/matcher_$seed.log
  echo "seed $seed done" >> $out/progress.log
done
