"""Randomised parity sweep: extractor vs oracle over random image sizes, pyramid parameters, thresholds, lapping
ranges, batch sizes and trig modes.  tests/test_gpu_sweeps.py runs a bounded fixed-seed slice of it on the GPU box;
alone: python tools/stress_parity.py [cases] [seed]"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import natural  # noqa: E402  (round 5: cuts of the two committed photographs, tests/natural.py)
import orb_slam3_detailed_comments_kor_amd as pkg  # noqa: E402
import orb_oracle_py as O  # noqa: E402

FIELDS = ("x", "y", "size", "angle", "response", "octave", "class_id")


def run(ncases=40, seed=7, max_side=(900, 1400), log=print):
    """Returns (valid configurations compared, list of mismatch descriptions)."""
    O.build()
    rng = np.random.default_rng(seed)
    bad, valid = [], 0
    for case in range(ncases):
        _one(case, rng, max_side, log, bad)
        valid += 1
    return valid - sum(1 for b in bad if b.startswith("rejected")), [b for b in bad if not b.startswith("rejected")]


def _one(case, rng, max_side, log, bad):
    if True:  # (one random configuration)
        H, W = int(rng.integers(200, max_side[0])), int(rng.integers(260, max_side[1]))
        nlev = int(rng.integers(3, 10))
        scale = float(np.float32(rng.choice([1.1, 1.2, 1.25, 1.3, 1.41, 1.5, 2.0])))
        nf = int(rng.integers(150, 3000))
        ini, mn = int(rng.integers(5, 60)), int(rng.integers(1, 30))
        lap = (int(rng.integers(0, W // 2)), int(rng.integers(W // 2, W + 200))) if rng.random() < 0.6 else (0, 0)
        trig = [pkg.binding.TRIG_LIBM, pkg.binding.TRIG_CR, pkg.binding.TRIG_LIBM_HOSTCHECK][case % 3]
        otrig = O.TRIG_CR if trig == pkg.binding.TRIG_CR else O.TRIG_LIBM
        # batches of >= 8 frames take the whole-images-per-XCD orders of K-PYR / K-FAST / K-DESC
        nb = int(rng.choice([1, 2, 3, 8, 9, 16])) if H * W < 450000 else int(rng.integers(1, 4))
        # content: rectangle frames, uniform noise, and (round 4) the kinds of synth.make_frame_kind -- blurred, saturated /
        # flat plateaus, 2-px checkerboard, fine sinusoids, pure ramp, quadrants of these
        kind = ("rects", "natural", "blurred", "noise", "plateaus", "checker2", "natural", "sinus", "mixed", "ramp", "rects", "noise")[case % 12]
        imgs = []
        for b in range(nb):
            if kind == "noise":
                imgs.append(rng.integers(0, 256, size=(H, W), dtype=np.uint8))
            elif kind == "natural":  # a photograph, cut / mirror-tiled to the case's size at a random offset and direction
                imgs.append(natural.random_frame(H, W, int(rng.integers(0, 1 << 30))))
            else:
                imgs.append(pkg.synth.make_frame_kind(H, W, int(rng.integers(0, 1 << 30)), kind))
        try:
            ex = pkg.ORBextractor(nf, scale, nlev, ini, mn, trig=trig)
        except Exception as e:
            log(case, "create failed", e)
            bad.append("rejected at create")
            return
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
            desc = "%s %s lev %d sf %s nF %d th %s lap %s trig %d batch %d n %s" % (kind, (H, W), nlev, scale, nf, (ini, mn), lap, trig, nb,
                                                                                 [len(o[1]) for o in outs])
            log(case, desc, "OK" if ok else "MISMATCH")
            if not ok:
                bad.append(desc)
        except pkg.OrbfeError as e:
            log(case, (H, W), "lev", nlev, "sf", scale, "rejected:", e)
            bad.append("rejected: unsupported configuration")
        finally:
            ex.close()


if __name__ == "__main__":
    n, bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    print("valid configurations:", n, "mismatches:", len(bad))
    sys.exit(1 if bad else 0)
