#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r05_qtwide
mkdir -p $out
cd $root
timeout -k 10 900 python3 -m pytest tests/test_gpu_extractor.py tests/test_gpu_configs.py tests/test_gpu_natural.py tests/test_gpu_content.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -6
for w in 1 0; do
ORBFE_QT_WIDE=$w python3 bench.py --config c4 --batch 8 --no-cpu-baseline --no-pcie --no-pipelined --no-cross > $out/c4b8_w$w.json 2> $out/c4b8_w$w.err || { tail -5 $out/c4b8_w$w.err; exit 1; }
ORBFE_QT_WIDE=$w python3 bench.py --config c4 --no-cpu-baseline --no-pcie --no-pipelined --no-cross > $out/c4b64_w$w.json 2> $out/c4b64_w$w.err || { tail -5 $out/c4b64_w$w.err; exit 1; }
python3 - <<PY
import json
for t in ("c4b8", "c4b64"):
    d = json.load(open("$out/%s_w$w.json" % t))
    print("wide=$w", t, "ms_per_step %.4f" % d["ms_per_step"], "one lane %.4f" % d["roofline"]["one_lane_ms_per_step"], {k: round(v * 1e3, 1) for k, v in d["roofline"]["stage_ms"].items()})
PY
done
ORBFE_QT_WIDE=1 python3 bench.py --config c5 --no-cpu-baseline > $out/c5_w1.json 2>/dev/null; ORBFE_QT_WIDE=0 python3 bench.py --config c5 --no-cpu-baseline > $out/c5_w0.json 2>/dev/null
python3 - <<PY
import json
for w in (1, 0):
    d = json.load(open("$out/c5_w%d.json" % w)); print("c5 wide=%d" % w, d["ms_per_step"])
PY
