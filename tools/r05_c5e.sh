#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
python3 - <<PY
import sys
sys.path.insert(0, "$root")
import bench
open("/tmp/frames1024.raw", "wb").write(bench.bench_frames(1024, 1024, 8).tobytes())
PY
show() { python3 -c "
import json,sys
d = json.loads(sys.stdin.read())
print('$1', {k: (v['ms_per_pair_p50'], v['extract_ms_p50']) for k, v in d.items() if isinstance(v, dict)})"; }
for k in 1 2 3; do tools/hostbench /tmp/frames1024.raw 1024 1024 8 1500 0 c5 | show c5; done
python3 bench.py --config c5 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k: v['ms_per_pair_p50'] for k, v in d['protocols_ms'].items()})"
