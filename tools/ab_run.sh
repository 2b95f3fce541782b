#!/bin/bash
# On the GPU box: bench every built variant (liborbfe_<name>.so) with the given extra env / bench args, print stage times.
#   tools/ab_run.sh "name1 name2 ..." [bench args]
root=${GRAFT_REPO_ROOT:-$PWD}
names=$1; shift
for n in $names; do
  lib=$root/orb_slam3_detailed_comments_kor_amd/liborbfe_$n.so
  [ "$n" = "default" ] && lib=$root/orb_slam3_detailed_comments_kor_amd/liborbfe.so
  ORBFE_LIB=$lib timeout -k 10 120 python bench.py --no-cpu-baseline --no-pcie --no-pipelined --no-cross --steps 200 "$@" > $root/gpurun_out/ab_$n.json 2>$root/gpurun_out/ab_$n.err
  python - <<PY
import json
try:
    j=json.loads(open("$root/gpurun_out/ab_$n.json").read().strip().splitlines()[-1])
    print("%-10s %.4f" % ("$n", j["ms_per_step"]), {k:round(v*1e3,1) for k,v in j["roofline"]["stage_ms"].items()})
except Exception as e:
    print("$n", "failed", e)
PY
done
