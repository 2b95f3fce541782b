#!/bin/bash
# A/B of the latency path at the drop-in boundary: tools/hostbench under two environments, same box, same call.
# usage: tools/ab_hostbench.sh "ENV_A=.." "ENV_B=.."   (results: gpurun_out/hb_a.json, hb_b.json)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python - <<'PY'
import bench
bench.bench_frames(480, 752, 64).tofile("/tmp/hb_frames.raw")
PY
for v in a b; do
  if [ $v = a ]; then E="$1"; else E="$2"; fi
  env $E timeout -k 10 200 tools/hostbench /tmp/hb_frames.raw 480 752 64 1000 0 > gpurun_out/hb_$v.json 2> gpurun_out/hb_$v.err
  python - "$v" "$E" <<'PY'
import json,sys
v=sys.argv[1]
p=json.loads(open("gpurun_out/hb_%s.json"%v).read().strip().splitlines()[-1])
print(v, 'fused pair', p.get('stereo_pair_fused'))
print(v, sys.argv[2], {k:p[k].get('ms_p50', p[k].get('ms_per_pair_p50', p[k].get('ms_per_batch'))) for k in ("single_pageable","single_pinned","single_pageable_autoreg","stereo_pair","stereo_pair_one_call","stereo_pair_one_call_pinned","batch_pageable","batch_pinned","batch_pipelined")})
PY
done
