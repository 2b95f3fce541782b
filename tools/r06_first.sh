#!/bin/bash
# round 6, first GPU pass: the new ComputeBoW tests + adapter, then the Python soak and more RCCL iterations of the C++ soak
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r06_first
mkdir -p $out
cd $root
timeout -k 10 600 python3 -m pytest tests/test_gpu_bow.py tests/test_gpu_vocabulary_adapter.py tests/test_gpu_keyframes.py tests/test_gpu_matcher.py -m gpu -x -q > $out/bow_tests.log 2>&1; rc=$?
tail -15 $out/bow_tests.log
[ $rc = 0 ] || exit $rc
bash tools/soak/run_soak.sh ${1:-420} 5 ${2:-420}
