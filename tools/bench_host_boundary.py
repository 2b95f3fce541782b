#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in boundary: orbfe_extract / orbfe_extract_batch with HOST pointers
(pageable numpy memory), i.e. what an unmodified Frame::ExtractORB sees.  Never used as bench.py's `value`
(which times frames resident in HBM); reported in DESIGN.md section 7.  One JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import orb_slam3_detailed_comments_kor_amd as pkg  # noqa: E402
from orb_slam3_detailed_comments_kor_amd.synth import make_frame  # noqa: E402


def main():
    rows, cols, B = 480, 752, 64
    frames = [make_frame(rows, cols, seed=100 + i) for i in range(B)]
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    for f in frames[:4]:
        ex(f)
    nkp = 0
    lat = []
    for rep in range(8): # 512 single-frame calls: mean and latency percentiles (ms/frame p50/p99)
        for f in frames:
            t0 = time.perf_counter()
            _, k, _ = ex(f)
            lat.append(time.perf_counter() - t0)
            if rep == 0:
                nkp += len(k)
    lat = np.array(lat)
    single = float(lat.mean())

    ex.extract_batch(frames)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        out = ex.extract_batch(frames)
    tb = (time.perf_counter() - t0) / reps
    print(json.dumps({"frame": "%dx%d" % (cols, rows), "single_call_ms": 1e3 * single,
                      "single_frames_per_s": 1.0 / single,
                      "single_call_ms_p50": 1e3 * float(np.percentile(lat, 50)),
                      "single_call_ms_p99": 1e3 * float(np.percentile(lat, 99)), "batch": B, "batch_call_ms": 1e3 * tb,
                      "batch_keypoints_per_s": nkp / tb, "keypoints_per_frame": nkp / B}))


if __name__ == "__main__":
    main()
