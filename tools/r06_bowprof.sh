#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r06_bowprof
rm -rf $out; mkdir -p $out
cd $root
python3 - <<PY > $out/frame.log 2>&1
import sys; sys.path.insert(0, "$root")
import orb_slam3_detailed_comments_kor_amd as pkg
pkg.synth.make_frame(480, 752, 77).tofile("$out/frame.raw")
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $root/tools/hostbench $out/frame.raw 480 752 1 1000 0 matcher > $out/hb.json 2> $out/hb.err || { tail -5 $out/hb.err; exit 1; }
rm -f $out/frame.raw
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$out/trace/*/*_kernel_stats.csv"))[-1]
for row in csv.DictReader(open(f)):
    n=row["Name"].split("(")[0].replace("void ","").replace("(anonymous namespace)::","")
    if any(k in n for k in ("bow","vocab","stage_in","cull")): print("%-40s calls %6s avg %9.1f ns min %s max %s"%(n[:40],row["Calls"],float(row["AverageNs"]),row["MinNs"],row["MaxNs"]))
PY
