#!/usr/bin/env python3
"""Phase timestamps of K-QT's level-0 workgroup (tuning; needs a library built with -DORBFE_QT_TIMING:
tools/ab_build.sh qtt "-DORBFE_QT_TIMING"; ORBFE_LIB=.../liborbfe_qtt.so python tools/qt_times.py [batch])."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import orb_slam3_detailed_comments_kor_amd as pkg  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
imgs = [pkg.synth.make_frame(480, 752, 1234 + i) for i in range(B)]
ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
t = np.zeros(64, np.uint64)
for _ in range(5):
    ex.extract_batch(imgs)
pkg.lib().orbfe_debug_qt_times(t.ctypes.data_as(C.c_void_p))  # (reading clears: the next batch is measured alone)
ex.extract_batch(imgs)
pkg.lib().orbfe_debug_qt_times(t.ctypes.data_as(C.c_void_p))
for lvl in range(8):
    v = int(t[30 + lvl])
    if v:
        print("level %d: slowest workgroup %.2f us (image %d)" % (lvl, (v >> 16) / 100.0, v & 0xFFFF))
    t[30 + lvl] = 0
t = t.astype(np.int64)
names = {0: "start", 1: "gather done", 2: "roots done", 41: "final phase begins", 59: "tree done", 60: "output written",
         5: "(gather) counts loaded", 6: "(gather) counts scanned", 10: "(roots) keys binned", 11: "(roots) checks done", 12: "(roots) check loops", 13: "(roots) check flags merged",
         20: "(round 1) cleared", 21: "(round 1) ranks", 22: "(round 1) kOf / par", 23: "(round 1) child histogram",
         24: "(round 1) growth scan + cut", 25: "(round 1) sidx scan", 26: "(round 1) expand", 27: "(output) best key per node", 28: "(output) keys + flags stored", 17: "(output) lapping range read", 14: "(round 1) rank keys loaded", 15: "(round 1) rank loop", 16: "(round 1) rank atomics", 29: "(output) tail slots cleared"}
print("first largest-first round: %d candidates, list size %d" % (int(t[61]) >> 32, int(t[61]) & 0xFFFFFFFF))
t[61] = 0
prev = t[0]
for k in sorted((k for k in range(61) if t[k] != 0), key=lambda k: (t[k], k)):
    print("%2d %-30s +%.2f us (at %.2f)" % (k, names.get(k, "pass / round"), (t[k] - prev) / 100.0, (t[k] - t[0]) / 100.0))
    prev = t[k]
