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
np.savez(sys.argv[1], **out)
""" % (ROOT, ROOT)


def run(lib, n, seeds):
    f = tempfile.mktemp(suffix=".npz")
    env = dict(os.environ)
    if lib:
        env["ORBFE_LIB"] = lib
    else:
        env.pop("ORBFE_LIB", None)
    r = subprocess.run([sys.executable, "-c", CHILD, f, str(seeds), str(n)], env=env, capture_output=True, text=True)
    print(lib or "liborbfe.so (default)", "|", r.stdout.strip(), r.stderr.strip()[-300:] if r.returncode else "")
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
