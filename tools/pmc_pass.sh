#!/bin/bash
# usage: tools/pmc_pass.sh <tag> "<counters>"   (run on the GPU box, from the repo root)
# One rocprofv3 counter pass over a short bench run; output CSVs under gpurun_out/pmc_<tag>/.
set -e
tag=$1; shift
ctrs="$1"; shift
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $root/gpurun_out/pmc_$tag -- python3 $root/bench.py --steps 4 --warmup 2 --settle 0 --lanes 1 --no-cpu-baseline --no-pipelined --no-pcie --no-cross "$@" > $root/gpurun_out/pmc_$tag.log 2>&1 || tail -5 $root/gpurun_out/pmc_$tag.log
