#!/bin/bash
# Round 6: everything under profiles/r06_* comes from this script (run on the GPU box through gpurun; the copies into profiles/
# are made in the build container from gpurun_out/r06_profiles/).
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r06_profiles
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# kernel trace of the step with ONE lane (a launch's duration is its throughput), then with the default lanes
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_one_lane -- python3 $root/bench.py --lanes 1 --no-cpu-baseline --no-pipelined --no-pcie --no-pmc --no-cross > $out/trace_one_lane.json 2> $out/trace_one_lane.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_default -- python3 $root/bench.py --no-cpu-baseline --no-pipelined --no-pcie --no-pmc > $out/trace_default.json 2> $out/trace_default.err || exit 1
cp $(ls -S $out/trace_one_lane/*/*_kernel_stats.csv | head -1) $out/kernel_stats.csv
cp $(ls -S $out/trace_default/*/*_kernel_stats.csv | head -1) $out/kernel_stats_default_lanes.csv
cd $root
# the bench lines (each with its in-run counter passes where the workload is the batched extractor, and its cpu_baseline)
python3 bench.py > $out/bench_752x480_b64.json 2> $out/bench_752x480_b64.err || exit 1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_752x480_b64_steps20_warmup5.json 2> $out/bench_drv.err || exit 1
python3 bench.py --config c4 > $out/bench_c4_1280x720_b64.json 2> $out/bench_c4.err || exit 1
python3 bench.py --config c4 --batch 8 > $out/bench_c4_1280x720_b8.json 2> $out/bench_c4b8.err || exit 1
python3 bench.py --config c3 > $out/bench_c3_stereo_pair.json 2> $out/bench_c3.err || exit 1
python3 bench.py --config c5 > $out/bench_c5_fisheye_pair.json 2> $out/bench_c5.err || exit 1
# the boundary from C++ under the kernel trace: which kernels a single host frame / a stereo frame / a batch runs, and for how long
python3 - <<PY > $out/frames.log 2>&1
import sys; sys.path.insert(0, "$root")
import numpy as np
import orb_slam3_detailed_comments_kor_amd as pkg
np.stack([pkg.synth.make_frame(480, 752, 77 + i) for i in range(64)]).tofile("$out/frames.raw")
PY
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_hostbench -- $root/tools/hostbench $out/frames.raw 480 752 64 1000 0 > $out/hostbench_traced.json 2> $out/hostbench_traced.err) || exit 1
cp $(ls -S $out/trace_hostbench/*/*_kernel_stats.csv | head -1) $out/hostbench_kernel_stats.csv
rm -f $out/frames.raw
# the matcher entry points from C++ (SearchByBoW x 64, the relocalisation chain, ...)
python3 - <<PY > $out/frame.log 2>&1
import sys; sys.path.insert(0, "$root")
import orb_slam3_detailed_comments_kor_amd as pkg
pkg.synth.make_frame(480, 752, 77).tofile("$out/frame.raw")
PY
tools/hostbench $out/frame.raw 480 752 1 1000 0 matcher > $out/matcher_hostbench.json 2> $out/matcher_hostbench.err || { tail -5 $out/matcher_hostbench.err; exit 1; }
rm -f $out/frame.raw
ls -la $out | head -40
