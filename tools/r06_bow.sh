#!/bin/bash
# ComputeBoW parity tests + the matcher entry points from C++ (the relocalisation chain among them)
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r06_bow
mkdir -p $out
cd $root
timeout -k 10 600 python3 -m pytest tests/test_gpu_bow.py tests/test_gpu_vocabulary_adapter.py tests/test_gpu_keyframes.py tests/test_gpu_multicam.py -m gpu -x -q > $out/tests.log 2>&1; rc=$?
tail -6 $out/tests.log
[ $rc = 0 ] || exit $rc
python3 - <<PY > $out/frame.log 2>&1
import sys; sys.path.insert(0, "$root")
import orb_slam3_detailed_comments_kor_amd as pkg
pkg.synth.make_frame(480, 752, 77).tofile("$out/frame.raw")
PY
tools/hostbench $out/frame.raw 480 752 1 1000 0 matcher > $out/matcher_hostbench.json 2> $out/matcher_hostbench.err || { tail -5 $out/matcher_hostbench.err; exit 1; }
rm -f $out/frame.raw
python3 - <<PY
import json
m=json.loads(open("$out/matcher_hostbench.json").read().strip().split("\n")[-1])
for k,v in m["calls"].items():
    if "bow" in k or "reloc" in k or "vocab" in k: print("  %-56s p50 %.4f mean %.4f"%(k,v["ms_p50"],v["ms_mean"]))
PY
