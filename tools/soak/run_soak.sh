#!/bin/bash
# tools/soak: the state of the round-5 abort, looped.  C++ only (RCCL, then the host transport), then the Python form.
# usage: tools/soak/run_soak.sh [seconds_cpp_rccl] [seconds_cpp_host] [seconds_py]
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r06_soak
mkdir -p $out
cd $out   # (a core file lands here)
ulimit -c unlimited
export LIBC_FATAL_STDERR_=1 NCCL_DEBUG=WARN
s1=${1:-360}; s2=${2:-200}; s3=${3:-300}
timeout -k 10 $((s1 + 120)) $root/tools/soak/mc_soak 0 $s1 > $out/cpp_rccl.log 2> $out/cpp_rccl.err; rc=$?
tail -2 $out/cpp_rccl.log; [ $rc = 0 ] || { echo "cpp rccl rc $rc"; tail -40 $out/cpp_rccl.err; exit $rc; }
timeout -k 10 $((s2 + 120)) $root/tools/soak/mc_soak 1 $s2 > $out/cpp_host.log 2> $out/cpp_host.err; rc=$?
tail -2 $out/cpp_host.log; [ $rc = 0 ] || { echo "cpp host rc $rc"; tail -40 $out/cpp_host.err; exit $rc; }
timeout -k 10 $((s3 + 180)) python3 $root/tools/soak/mc_soak.py $s3 rccl > $out/py_rccl.log 2> $out/py_rccl.err; rc=$?
tail -2 $out/py_rccl.log; [ $rc = 0 ] || { echo "py rccl rc $rc"; tail -60 $out/py_rccl.err; exit $rc; }
ls -la $out
