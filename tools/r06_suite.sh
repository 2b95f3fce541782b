#!/bin/bash
# the whole GPU suite, then the default bench line (with its in-run counter passes) and the driver's short form
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r06_suite
mkdir -p $out
cd $root
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; rc=$?
tail -8 $out/gpu_tests.log
[ $rc = 0 ] || exit $rc
timeout -k 10 500 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; rc=$?
tail -c 600 $out/bench_default.err; python3 - <<PY
import json
j=json.loads(open("$out/bench_default.json").read().strip().split("\n")[-1])
r=j["roofline"]
print("value %.1f M kp/s  ms_per_step %.4f  dom %s frac %.3f issue_frac %s traffic %s t/alg %s step.frac %.3f" % (j["value"]/1e6, j["ms_per_step"], r["kernel"], r["frac"], r.get("issue_frac"), r.get("traffic"), r.get("traffic_over_algorithmic"), r["step"]["frac"]))
print(r.get("traffic_source"))
print({k:(round(v["issue_frac"],3), round(v["cycles_per_instruction"],2)) for k,v in (r.get("issue") or {}).get("kernels",{}).items()})
print("stage_ms", r["stage_ms"])
cb=j.get("cpu_baseline",{}); print("cpu", cb.get("value"), cb.get("one_thread",{}).get("stage_ms"), (cb.get("scalar_port") or {}).get("value"))
print("vs_cpu", j.get("vs_cpu")); print("single", j.get("single_frame")); print("pcie single", {k:v for k,v in (j.get("pcie_inclusive") or {}).items() if "single" in k})
PY
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err; rc=$?
python3 -c "
import json;j=json.loads(open('$out/bench_driver.json').read().strip().split('\n')[-1]);print('driver form: value %.1f M ms %.4f' % (j['value']/1e6, j['ms_per_step']))"
exit $rc
