#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_pairs
mkdir -p $out
cd $root
timeout -k 10 900 python3 -m pytest tests/test_gpu_hostpath.py tests/test_gpu_lanes.py tests/test_gpu_multicam.py -m gpu -x -q > $out/pytest.log 2>&1
rc=$?
tail -5 $out/pytest.log
[ $rc = 0 ] || exit $rc
python3 bench.py --config c3 --no-cpu-baseline > $out/c3.json 2> $out/c3.err || { tail -5 $out/c3.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("$out/c3.json"))
print(json.dumps(d["protocols_ms"], indent=1))
PY
for l in 1 2 3; do
ORBFE_BENCH_FORCE_DIST=1 python3 bench.py --config c4 --batch 8 --lanes $l --no-cpu-baseline --no-pcie --no-pipelined > $out/c4b8_dist_l$l.json 2> $out/c4b8_dist_l$l.err || { tail -5 $out/c4b8_dist_l$l.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("$out/c4b8_dist_l$l.json"))
print("c4b8 one-rank RCCL lanes $l ms_per_step=%.4f cross=%s" % (d["ms_per_step"], (d.get("cross_camera") or {}).get("ms_per_step")), flush=True)
PY
done
