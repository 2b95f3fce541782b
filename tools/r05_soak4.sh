#!/bin/bash
# host heap checking (glibc MALLOC_CHECK_=3 through libc_malloc_debug.so, which glibc >= 2.34 needs for it: a guard byte behind every block, verified at free) under the four test files whose
# process aborted once in tools/r05_soak.sh: an overflow of a caller-side array by the library or the binding shows up at its free
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_soak
mkdir -p $out
cd $root
export LD_PRELOAD=/lib/x86_64-linux-gnu/libc_malloc_debug.so.0 LIBC_FATAL_STDERR_=1 MALLOC_CHECK_=3 MALLOC_PERTURB_=165 PYTHONFAULTHANDLER=1 PYTHONMALLOC=malloc
timeout -k 10 900 python3 -X faulthandler -m pytest tests/test_gpu_lanes.py tests/test_gpu_hostpath.py tests/test_gpu_multicam.py tests/test_gpu_keyframes.py -m gpu -x -q -v > $out/mcheck.log 2> $out/mcheck.err
rc=$?
tail -3 $out/mcheck.log
grep -v "^  File\|amdgpu.ids" $out/mcheck.err | tail -20
exit $rc
