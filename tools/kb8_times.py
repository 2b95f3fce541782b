"""Where a thread of KannalaBrandt8::TriangulateMatches spends its time (a library built with -DORBFE_KB8_TIMING:
tools/ab_build.sh kbt "-DORBFE_KB8_TIMING"; ORBFE_LIB=.../liborbfe_kbt.so python tools/kb8_times.py).  Tuning only."""
import ctypes as C
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orb_slam3_detailed_comments_kor_amd as pkg
from matcher_inputs import stereo_fisheye_inputs

I = stereo_fisheye_inputs(3, 1500, 1500)
args = (I["descL"], I["kpL"], I["octL"], I["descR"], I["kpR"], I["octR"], I["P1"], I["P2"], I["Rlr"], I["tlr"], I["sig"])
for _ in range(20):
    n = pkg.stereo_fisheye_matches(*args)[0]
t = np.zeros(8, np.uint64)
L = pkg.lib()
L.orbfe_debug_kb8_times.argtypes = [C.c_void_p]
assert L.orbfe_debug_kb8_times(t.ctypes.data_as(C.c_void_p)) == 0
tot = float(t[:4].sum())
print("matches per call", n)
for k, name in enumerate(("two unprojections", "parallax + A", "SVD", "two projections")):
    print("%-20s %5.1f %%  (%.0f ticks)" % (name, 100.0 * float(t[k]) / max(tot, 1), float(t[k])))
