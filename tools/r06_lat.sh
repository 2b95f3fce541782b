#!/bin/bash
# after the latency-path changes: the host-path + bow + keyframe tests, then hostbench (single frame, stereo, matcher) and its kernel trace
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r06_lat
rm -rf $out; mkdir -p $out
cd $root
timeout -k 10 900 python3 -m pytest tests/test_gpu_hostpath.py tests/test_gpu_bow.py tests/test_gpu_vocabulary_adapter.py tests/test_gpu_adapter.py tests/test_gpu_configs.py tests/test_gpu_extractor.py -m gpu -x -q > $out/tests.log 2>&1; rc=$?
tail -6 $out/tests.log
[ $rc = 0 ] || exit $rc
python3 - <<PY > $out/frame.log 2>&1
import sys; sys.path.insert(0, "$root")
import numpy as np
import orb_slam3_detailed_comments_kor_amd as pkg
np.stack([pkg.synth.make_frame(480, 752, 77 + i) for i in range(64)]).tofile("$out/frames.raw")
PY
tools/hostbench $out/frames.raw 480 752 64 1000 0 > $out/hostbench.json 2> $out/hostbench.err || { tail -5 $out/hostbench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $root/tools/hostbench $out/frames.raw 480 752 1 1000 0 matcher > $out/matcher.json 2> $out/matcher.err || { tail -5 $out/matcher.err; exit 1; }
rm -f $out/frames.raw
python3 - <<PY
import csv,glob,json
j=json.loads(open("$out/hostbench.json").read().strip().split("\n")[-1])
for k,v in j.items():
    if isinstance(v,dict) and ("ms_mean" in v or "ms_p50" in v): print("  %-40s %s"%(k,{a:b for a,b in v.items() if a.startswith("ms")}))
m=json.loads(open("$out/matcher.json").read().strip().split("\n")[-1])
for k,v in m["calls"].items():
    if "bow" in k or "reloc" in k: print("  %-56s p50 %.4f"%(k,v["ms_p50"]))
f=sorted(glob.glob("$out/trace/*/*_kernel_stats.csv"))[-1]
for row in csv.DictReader(open(f)):
    n=row["Name"].replace("void ","").replace("(anonymous namespace)::","").split("(")[0]
    print("  %-44s calls %6s avg %9.1f ns"%(n[:44],row["Calls"],float(row["AverageNs"])))
PY
