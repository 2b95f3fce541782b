#!/bin/bash
# Experiments on the triangulation's Jacobi SVD (round 5; variants built with tools/ab_build.sh or into build_ab/<name>/liborbfe.so):
#   python3 tools/kb8_ab.py <other liborbfe.so> 30000 6        two builds bit for bit + the time of the stereo-fisheye call
#   LD_LIBRARY_PATH=build_ab/<name> tools/hostbench <frames> 1024 1024 8 1500 0 c5     BASELINE configs[4] under a variant
# Results (one MI355X): sweeps capped at 3 (-DORBFE_SVD_MAXIT=3, wrong results) 0.300 -> 0.229 ms per fisheye stereo frame;
# fewer instructions per rotation: no change; two rotations at a time: 0.317 -> 0.366.
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
python3 - <<PY
import sys
sys.path.insert(0, "$root")
import bench
open("/tmp/frames1024.raw", "wb").write(bench.bench_frames(1024, 1024, 8).tobytes())
PY
for v in "$@"; do
  echo "== $v"
  LD_LIBRARY_PATH=$root/build_ab/$v:$LD_LIBRARY_PATH tools/hostbench /tmp/frames1024.raw 1024 1024 8 1500 0 c5 | cut -c1-1500
done
