#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_knn
mkdir -p $out
cd $root
timeout -k 10 600 python3 -m pytest tests/test_gpu_resident.py tests/test_gpu_multicam.py tests/test_gpu_matcher.py -m gpu -x -q -k "knn or frames or multicam or resident" > $out/pytest.log 2>&1
rc=$?
tail -15 $out/pytest.log
[ $rc = 0 ] || exit $rc
for m in 1 0; do
ORBFE_KNN2_MFMA=$m python3 bench.py --no-cpu-baseline --no-pcie --no-pipelined --steps 100 > $out/cross_m$m.json 2> $out/cross_m$m.err || { tail -5 $out/cross_m$m.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("$out/cross_m$m.json"))
print("mfma=$m", json.dumps(d.get("cross_camera")))
PY
done
