#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r06_bvt
rm -rf $out; mkdir -p $out
cd $root
python3 - <<PY > $out/frame.log 2>&1
import sys; sys.path.insert(0, "$root")
import orb_slam3_detailed_comments_kor_amd as pkg
pkg.synth.make_frame(480, 752, 77).tofile("$out/frame.raw")
PY
tools/hostbench_bvt $out/frame.raw 480 752 1 1000 0 matcher > $out/hb.json 2> $out/hb.err
grep "k_bow_rank_fold" $out/hb.err
rm -f $out/frame.raw
