#!/usr/bin/env python3
"""Phase durations of K-FAST summed over all workgroups of one batch (tuning; needs a library built with
-DORBFE_FAST_TIMING: tools/ab_build.sh ft "-DORBFE_FAST_TIMING"; ORBFE_LIB=.../liborbfe_ft.so python tools/fast_times.py [batch])."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import orb_slam3_detailed_comments_kor_amd as pkg  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
imgs = [pkg.synth.make_frame(480, 752, 1234 + i) for i in range(B)]
ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
t = np.zeros(16, np.uint64)
for _ in range(5):
    ex.extract_batch(imgs)
pkg.lib().orbfe_debug_fast_times(t.ctypes.data_as(C.c_void_p))  # (reading clears: the next batch is measured alone)
ex.extract_batch(imgs)
pkg.lib().orbfe_debug_fast_times(t.ctypes.data_as(C.c_void_p))
n = int(t[15])
names = ["cell record", "staging issue+wait", "barrier", "phase A", "barrier", "phase B (score)", "barrier", "phase C (NMS)",
         "barrier", "scan + output"]
if os.environ.get("ORBFE_FAST_RUNS", "0") not in ("0", ""):  # round 4's kernel: a workgroup per run of cells, two more phases
    names = ["run record", "staging issue+wait", "barrier", "phase A", "barrier", "phase B (score + corner queue)", "barrier",
             "NMS over the corners", "barrier", "per-cell ranks", "barrier", "ranked output"]
tot = 0.0
for k, nm in enumerate(names):
    us = float(t[k]) / 100.0 / max(n, 1)
    tot += us
    print("%-20s %.3f us per workgroup" % (nm, us))
print("workgroups %d, mean life %.3f us" % (n, tot))
