#!/bin/bash
# C5: the pair of 1024 x 1024 pageable images staged and uploaded image by image (A/B: ORBFE_PAIR_PER_IMAGE=0)
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_c5
mkdir -p $out
cd $root
timeout -k 10 600 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_hostpath.py -m gpu -x -q > $out/pytest.log 2>&1
rc=$?
tail -3 $out/pytest.log
[ $rc = 0 ] || exit $rc
python3 - <<PY
import sys
sys.path.insert(0, "$root")
import bench
open("/tmp/frames1024.raw", "wb").write(bench.bench_frames(1024, 1024, 8).tobytes())
PY
show() { python3 -c "
import json,sys
d = json.loads(sys.stdin.read())
print('$1', {k: (v['ms_per_pair_p50'], v['extract_ms_p50']) for k, v in d.items() if isinstance(v, dict)})"; }
for k in 1 2 3; do
  ORBFE_PAIR_PER_IMAGE=0 tools/hostbench /tmp/frames1024.raw 1024 1024 8 1500 0 c5 | show together
  tools/hostbench /tmp/frames1024.raw 1024 1024 8 1500 0 c5 | show perimage
done
