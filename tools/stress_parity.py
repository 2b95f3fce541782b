"""Randomised parity sweep (not part of the test-suite): extractor vs oracle over random image sizes, pyramid
parameters, thresholds, lapping ranges, batch sizes and trig modes.  Usage: python tools/stress_parity.py [cases] [seed]"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orb_slam3_detailed_comments_kor_amd as pkg  # noqa: E402
import orb_oracle_py as O  # noqa: E402

O.build()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
FIELDS = ("x", "y", "size", "angle", "response", "octave", "class_id")
bad = 0
for case in range(ncases):
    H, W = int(rng.integers(200, 900)), int(rng.integers(260, 1400))
    nlev = int(rng.integers(3, 10))
    scale = float(np.float32(rng.choice([1.1, 1.2, 1.25, 1.3, 1.41, 1.5, 2.0])))
    nf = int(rng.integers(150, 3000))
    ini, mn = int(rng.integers(5, 60)), int(rng.integers(1, 30))
    lap = (int(rng.integers(0, W // 2)), int(rng.integers(W // 2, W + 200))) if rng.random() < 0.6 else (0, 0)
    trig = [pkg.binding.TRIG_LIBM, pkg.binding.TRIG_CR, pkg.binding.TRIG_LIBM_HOSTCHECK][case % 3]
    otrig = O.TRIG_CR if trig == pkg.binding.TRIG_CR else O.TRIG_LIBM
    # batches of >= 8 frames take the whole-images-per-XCD orders of K-PYR / K-FAST / K-DESC
    nb = int(rng.choice([1, 2, 3, 8, 9, 16])) if H * W < 450000 else int(rng.integers(1, 4))
    kind = case % 4
    imgs = []
    for b in range(nb):
        if kind == 3:
            imgs.append(rng.integers(0, 256, size=(H, W), dtype=np.uint8))
        else:
            imgs.append(pkg.synth.make_frame(H, W, int(rng.integers(0, 1 << 30))))
    try:
        ex = pkg.ORBextractor(nf, scale, nlev, ini, mn, trig=trig)
    except Exception as e:
        print(case, "create failed", e)
        continue
    try:
        ref = O.Extractor(nf, scale, nlev, ini, mn, trig=otrig)
        if nb == 1:
            outs = [ex(imgs[0], lap)]
        else:
            outs = ex.extract_batch(imgs, [lap] * nb)
        ok = True
        for b in range(nb):
            mono, kps, desc = outs[b]
            rmono, rkps, rdesc = ref.extract(imgs[b], lap, cap=4 * nf + 400)
            same = mono == rmono and len(kps) == len(rkps) and np.array_equal(desc, rdesc) and \
                all(np.array_equal(kps[f], rkps[f]) for f in FIELDS)
            ok &= bool(same)
        print(case, (H, W), "lev", nlev, "sf", scale, "nF", nf, "th", (ini, mn), "lap", lap, "trig", trig, "batch", nb,
              "n", [len(o[1]) for o in outs], "OK" if ok else "MISMATCH")
        bad += 0 if ok else 1
    except pkg.OrbfeError as e:
        print(case, (H, W), "lev", nlev, "sf", scale, "rejected:", e)
    finally:
        ex.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
