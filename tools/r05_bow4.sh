#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_bow
mkdir -p $out
cd $root
timeout -k 10 600 python3 -m pytest tests/test_gpu_keyframes.py -m gpu -x -q > $out/pytest.log 2>&1
rc=$?
tail -3 $out/pytest.log
[ $rc = 0 ] || exit $rc
python3 - <<PY
import sys
sys.path.insert(0, "$root")
import bench
open("/tmp/frames.raw", "wb").write(bench.bench_frames(480, 752, 8).tobytes())
PY
for mode in 0 1; do
  ORBFE_BOW_DEVNODES=$mode timeout -k 10 300 tools/hostbench /tmp/frames.raw 480 752 8 1200 0 matcher > $out/matcher_dev$mode.json 2> $out/matcher_dev$mode.err || tail -3 $out/matcher_dev$mode.err
  python3 -c "
import json; d = json.load(open('$out/matcher_dev$mode.json'))['calls']
print('devnodes $mode', {k: v['ms_p50'] for k, v in d.items() if 'bow' in k})"
done
bash tools/r05_bow3.sh
