#!/bin/bash
# round 5, first GPU call: (1) what overlap of WHOLE batches buys in the < 16-frame regime with what exists
# (bench.py --contexts N = N extractor contexts round-robin); (2) rocprof + PMC of the matcher kernels as they are now.
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_first
mkdir -p $out
cd $root
common="--config c4 --batch 8 --steps 300 --warmup 30 --no-cpu-baseline --no-pcie --no-cross --no-pipelined"
for q in default 8; do
  for c in 1 2 3 4; do
    for l in 1 2; do
      if [ $q = 8 ]; then export GPU_MAX_HW_QUEUES=8; else unset GPU_MAX_HW_QUEUES; fi
      python3 bench.py $common --contexts $c --lanes $l > $out/c4b8_q${q}_c${c}_l${l}.json 2> $out/c4b8_q${q}_c${c}_l${l}.err || exit 1
      python3 - <<PY
import json
d = json.load(open("$out/c4b8_q${q}_c${c}_l${l}.json"))
print("c4 b8 queues=$q contexts=$c lanes=$l ms_per_step=%.4f" % d["ms_per_step"], flush=True)
PY
    done
  done
done
unset GPU_MAX_HW_QUEUES
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/matcher_trace -- python3 $root/tools/bench_matcher.py > $out/matcher_trace.log 2>&1 || exit 1
echo matcher trace done
for set in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sqa:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  tag=${set%%:*}; ctrs=${set#*:}
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/matcher_pmc_$tag -- python3 $root/tools/bench_matcher.py > $out/matcher_pmc_$tag.log 2>&1 || { tail -5 $out/matcher_pmc_$tag.log; exit 1; }
  echo matcher pmc $tag done
done
