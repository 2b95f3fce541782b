#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
timeout -k 10 600 python3 -m pytest tests/test_gpu_keyframes.py tests/test_gpu_matcher.py -m gpu -x -q 2>&1 | tail -12
python3 - <<PY
import sys
sys.path.insert(0, "$root")
import bench
open("/tmp/frames.raw", "wb").write(bench.bench_frames(480, 752, 8).tobytes())
PY
tools/hostbench /tmp/frames.raw 480 752 8 1200 0 matcher 2>&1 | tr ',' '\n' | grep -A3 "search_projection_batch64\|search_bow_batch64" | head -40
