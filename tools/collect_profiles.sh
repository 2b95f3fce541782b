#!/bin/bash
# Run on the GPU box from the repo root (via gpurun).  Produces under gpurun_out/:
#   prof_trace/   rocprofv3 --kernel-trace --stats of the default bench command
#   pmc_*/        separate counter passes (FETCH_SIZE, WRITE_SIZE, SQ sets), never combined with tracing domains
#   calib/        FETCH_SIZE calibration on known byte counts
set -x
root=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $root/gpurun_out
rm -rf $root/gpurun_out/pmc_* $root/gpurun_out/prof_trace $root/gpurun_out/prof_trace_default $root/gpurun_out/calib
cd /tmp && export TMPDIR=/tmp
# (one lane: with two lanes the kernels of the two half-batches overlap and a launch's wall duration is not its throughput;
# the roofline object of the bench line is measured with one lane too)
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_trace -- python3 $root/bench.py --lanes 1 --no-cpu-baseline --no-pipelined --no-pcie > $root/gpurun_out/prof_trace.log 2>&1
# ... and the default command (two lanes in the timed region, one lane in the roofline region), for the record
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_trace_default -- python3 $root/bench.py --no-cpu-baseline --no-pipelined --no-pcie --no-cross > $root/gpurun_out/prof_trace_default.log 2>&1
cd $root
tools/pmc_pass.sh fetch "FETCH_SIZE"
tools/pmc_pass.sh write "WRITE_SIZE"
tools/pmc_pass.sh sqa "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
tools/pmc_pass.sh sqb "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o /tmp/pmc_calib tools/pmc_calib.hip
cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/calib -- /tmp/pmc_calib > $root/gpurun_out/calib.log 2>&1
cd $root
python3 tools/pmc_summary.py gpurun_out > gpurun_out/pmc_summary.json
