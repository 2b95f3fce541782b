#!/bin/bash
# the fixed-point exit of the Jacobi SVD against a build without it (build_ab/nofix): bit for bit, tests, hostbench c5
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
timeout -k 10 600 python3 tools/kb8_ab.py $root/build_ab/nofix/liborbfe.so 30000 6 || exit 1
timeout -k 10 600 python3 -m pytest tests/test_gpu_matcher.py tests/test_gpu_configs.py tests/test_gpu_matcher_adapter.py -m gpu -x -q 2>&1 | tail -2
python3 - <<PY
import sys
sys.path.insert(0, "$root")
import bench
open("/tmp/frames1024.raw", "wb").write(bench.bench_frames(1024, 1024, 8).tobytes())
PY
show() { python3 -c "
import json,sys
d = json.loads(sys.stdin.read())
print('$1', d['matches_per_pair'], {k: (v['ms_per_pair_p50'], v['extract_ms_p50']) for k, v in d.items() if isinstance(v, dict)})"; }
for k in 1 2; do
  LD_LIBRARY_PATH=$root/build_ab/nofix:$LD_LIBRARY_PATH tools/hostbench /tmp/frames1024.raw 1024 1024 8 1500 0 c5 | show nofix
  tools/hostbench /tmp/frames1024.raw 1024 1024 8 1500 0 c5 | show fixpoint
done
