import ctypes as C, numpy as np, sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import orb_slam3_detailed_comments_kor_amd as pkg
L = pkg.lib()
ex = pkg.ORBextractor(1200, 1.2, 8, 20, 7)
a = pkg.synth.make_frame(480, 752, 1)
b = np.roll(a, -12, axis=1)
for _ in range(30):
    m = pkg.binding.extract_stereo_pair(ex, a, b, 0.11, 47.9)
t = np.zeros(8, np.uint64)
L.orbfe_debug_stereo_times.argtypes = [C.c_void_p]
print("rc", L.orbfe_debug_stereo_times(t.ctypes.data_as(C.c_void_p)), t)
n = max(int(t[7]), 1)
print("mean us: staged %.2f scanned %.2f scored %.2f sad %.2f stored %.2f" % tuple(float(t[k]) * 0.01 / n for k in range(1, 6)))
