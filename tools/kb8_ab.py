"""Two builds of the library on the same triangulations, bit for bit, and the time of orbfe_stereo_fisheye_matches under each
(tuning: tools/ab_build.sh <name> "<flags>", then python tools/kb8_ab.py <liborbfe_name.so> [pairs per seed] [seeds]).  Each
library runs in a child process (ORBFE_LIB).  Round 5 used it for a Jacobi SVD with v_fma_f64 for the exact float x float
products and a v_sqrt_f64 filter in front of the convergence test's correctly rounded square root: identical on 180 000
triangulations, 0.0820 against 0.0824 ms per call -- not kept."""
import os
import subprocess
import sys
import tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import orb_slam3_detailed_comments_kor_amd as pkg
from matcher_inputs import kb8_pairs, stereo_fisheye_inputs
import time
out = {}
for seed in range(int(sys.argv[2])):
    G = kb8_pairs(1000 + seed, int(sys.argv[3]))
    z, X = pkg.kb8_triangulate(G["P1"], G["P2"], G["kp1"], G["kp2"], G["R12"], G["t12"], G["sigma1"], G["sigma2"])
    out["z%%d" %% seed] = z; out["X%%d" %% seed] = X
I = stereo_fisheye_inputs(3, 1500, 1500)
args = (I["descL"], I["kpL"], I["octL"], I["descR"], I["kpR"], I["octR"], I["P1"], I["P2"], I["Rlr"], I["tlr"], I["sig"])
for _ in range(30): r = pkg.stereo_fisheye_matches(*args)
t = time.perf_counter()
for _ in range(300): r = pkg.stereo_fisheye_matches(*args)
print("ms per stereo_fisheye_matches (1500 x 1500): %%.4f" %% (1e3 * (time.perf_counter() - t) / 300))
out["l2r"], out["dep"], out["p3d"] = r[1], r[3], r[4]
# ... and what tools/hostbench c5 matches: a 1024 x 1024 frame against itself shifted by 40 px (matrices that never meet the
# convergence test), TUM-VI parameters scaled to the image, a 10-cm baseline
sys.path.insert(0, %r)
import bench
fr = bench.bench_frames(1024, 1024, 2)
ex = pkg.ORBextractor(1500, 1.2, 8, 20, 7)
sc = 1024 / 512.0
P = np.array([190.978477 * sc, 190.973307 * sc, 254.931706 * sc, 256.897442 * sc, 0.003482389, 0.000715034, -0.002053236, 0.000202937], np.float32)
sig = (1.2 ** np.arange(8)).astype(np.float32) ** 2
for k in range(2):
    L = fr[k]; R = np.roll(L, -40, axis=1)
    (mL, kL, dL), (mR, kR, dR) = ex.extract_batch([L, R], [(0, 1023)] * 2)
    xyL = np.stack([kL["x"][mL:], kL["y"][mL:]], 1); xyR = np.stack([kR["x"][mR:], kR["y"][mR:]], 1)
    a2 = (dL[mL:], xyL, kL["octave"][mL:], dR[mR:], xyR, kR["octave"][mR:], P, P, np.eye(3, dtype=np.float32), np.array([0.101, 0, 0], np.float32), sig)
    r = pkg.stereo_fisheye_matches(*a2)
    for _ in range(20): pkg.stereo_fisheye_matches(*a2)
    t = time.perf_counter()
    for _ in range(100): pkg.stereo_fisheye_matches(*a2)
    print("shifted pair %%d: %%d x %%d keypoints, %%d matches, %%.4f ms per call" %% (k, len(xyL), len(xyR), r[0], 1e3 * (time.perf_counter() - t) / 100))
    out["s_l2r%%d" %% k], out["s_dep%%d" %% k], out["s_p3d%%d" %% k] = r[1], r[3], r[4]
ex.close()
np.savez(sys.argv[1], **out)
""" % (ROOT, ROOT, ROOT)


def run(lib, n, seeds):
    f = tempfile.mktemp(suffix=".npz")
    env = dict(os.environ)
    if lib:
        env["ORBFE_LIB"] = lib
    else:
        env.pop("ORBFE_LIB", None)
    r = subprocess.run([sys.executable, "-c", CHILD, f, str(seeds), str(n)], env=env, capture_output=True, text=True)
    print(lib or "liborbfe.so (default)", "|", r.stdout.strip().replace("\n", " | "), r.stderr.strip()[-600:] if r.returncode else "")
    if r.returncode:
        sys.exit(1)
    return dict(np.load(f))


if __name__ == "__main__":
    other = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
    seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    a, b = run(None, n, seeds), run(other, n, seeds)
    bad = [k for k in a if not np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8))]
    tot = sum(len(a["z%d" % s]) for s in range(seeds))
    print("triangulations compared:", tot, "accepted:", int(sum((a["z%d" % s] > 0).sum() for s in range(seeds))), "arrays that differ:", bad)
    sys.exit(1 if bad else 0)
