"""Randomised parity sweep of the matcher entry points vs the oracle.  tests/test_gpu_sweeps.py runs a bounded
fixed-seed slice of it on the GPU box; alone: python tools/stress_matcher.py [cases] [seed]"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orb_slam3_detailed_comments_kor_amd as pkg  # noqa: E402
import orb_oracle_py as O  # noqa: E402
import matcher_inputs as MI  # noqa: E402

def run(ncases=60, seed=3, scale=1.0, log=print):
    """Returns the list of mismatch descriptions.  scale < 1 shrinks the random problem sizes (test-suite slice)."""
    O.build()
    rng = np.random.default_rng(seed)
    bad = []

    def size(lo, hi):
        return int(rng.integers(lo, max(lo + 1, int(lo + (hi - lo) * scale))))

    for case in range(ncases):
        kind = case % 6
        seed = int(rng.integers(0, 1 << 30))
        ok = True
        desc = ""
        if kind == 0:    # SearchByProjection local map / last frame
            mode = int(rng.integers(0, 2))
            kw = dict(seed=seed, mode=mode, n=size(50, 3000), nq=size(1, 2500),
                      th=float(rng.choice([1.0, 3.0, 7.0, 15.0])), stereo=bool(rng.integers(0, 2)),
                      check_orientation=bool(mode and rng.integers(0, 2)), taken_frac=float(rng.uniform(0, 0.4)),
                      nnratio=float(rng.choice([0.6, 0.8, 0.9])))
            if rng.random() < 0.3:
                kw.update(stereo=False, Nleft=int(kw["n"] * rng.uniform(0.3, 0.7)), partners=bool(mode == 0 and rng.integers(0, 2)))
            elif rng.random() < 0.3:
                kw["blocks"] = float(rng.uniform(0.2, 0.9))
            pr = MI.projection_problem(**kw)
            a, b = O.search_projection(pr), pkg.search_projection(pr)
            ok = a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
            desc = "proj %s -> %d" % (kw, a[0])
        elif kind == 1:  # Fuse / Sim3 family
            loop = str(rng.choice(["sim3_projection", "fuse", "fuse_sim3", "search_by_sim3"]))
            kw = dict(seed=seed, mode=1, n=size(50, 2500), nq=size(1, 2000),
                      th=float(rng.choice([2.5, 3.0, 4.0, 7.5])), loop=loop, stereo=bool(loop == "fuse" and rng.integers(0, 2)))
            if loop == "fuse" and not kw["stereo"] and rng.random() < 0.4:
                kw["Nleft"] = int(kw["n"] * rng.uniform(0.3, 0.7))
            pr = MI.projection_problem(**kw)
            a, b = O.search_projection(pr), pkg.search_projection(pr)
            ok = a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
            desc = "%s %s -> %d" % (loop, {k: kw[k] for k in ("n", "nq", "th")}, a[0])
        elif kind == 2:  # SearchForInitialization
            kw = dict(seed=seed, n1=size(20, 3000), n2=size(2, 3000), window=int(rng.choice([30, 100, 200])),
                      nnratio=float(rng.choice([0.7, 0.9, 1.0])), check_orientation=bool(rng.integers(0, 2)), crowd=bool(rng.integers(0, 2)))
            pr = MI.initialization_problem(**kw)
            a, b = O.search_initialization(pr), pkg.search_initialization(pr)
            ok = a[0] == b[0] and np.array_equal(a[1], b[1])
            desc = "init %s -> %d" % (kw, a[0])
        elif kind == 3:  # SearchByBoW both variants
            n1, n2 = size(30, 2000), size(30, 2000)
            d1, d2, a1, a2 = MI.descriptor_sets(n1, n2, seed % 100000)
            fv1, fv2 = MI.feature_vectors(d1, d2, seed % 1000, int(rng.integers(3, 11)), 2)
            m1 = (rng.uniform(size=n1) < 0.6).astype(np.uint8)
            m2 = (rng.uniform(size=n2) < 0.6).astype(np.uint8)
            ori = bool(rng.integers(0, 2))
            ratio = float(rng.choice([0.6, 0.7, 0.9]))
            a = O.search_bow_kf_f(d1, m1, a1, fv1, d2, a2, fv2, -1, ratio, ori)
            b = pkg.search_bow(d1, m1, a1, fv1, d2, None, a2, fv2, 0, ratio, ori)
            c = O.search_bow_kf_kf(d1, m1, a1, fv1, d2, m2, a2, fv2, -1, -1, ratio, ori)
            d = pkg.search_bow(d1, m1, a1, fv1, d2, m2, a2, fv2, 1, ratio, ori)
            ok = a[0] == b[0] and np.array_equal(a[1], b[1]) and c[0] == d[0] and np.array_equal(c[1], d[1])
            desc = "bow n1=%d n2=%d -> %d / %d" % (n1, n2, a[0], c[0])
        elif kind == 4:  # SearchForTriangulation_ pinhole
            n1, n2 = size(50, 2000), size(50, 2000)
            I = MI.tri_inputs(n1, n2, seed % 100000)
            os_, co = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            args = (I["d1"], I["has1"], I["kp1"], I["a1"], I["oct1"], I["u1"], I["fv1"], I["d2"], I["has2"], I["kp2"], I["a2"],
                    I["oct2"], I["u2"], I["fv2"], I["F12"], I["ep"], I["sf"], I["sig"], os_, co, True)
            a, b = O.search_triangulation(*args), pkg.search_triangulation(*args)
            ok = np.array_equal(a, b)
            desc = "tri n1=%d n2=%d stereo=%s coarse=%s -> %d" % (n1, n2, os_, co, len(a))
        else:            # knn-2 and all-pairs distances, ragged sizes
            nq, nt = size(1, 1800), size(1, 1800)
            Q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
            T = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
            T[rng.integers(0, nt, max(nt // 10, 1))] = Q[rng.integers(0, nq, max(nt // 10, 1))]
            a, b = O.bfknn2(Q, T), pkg.bfknn2(Q, T)
            ok = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(O.hamming_matrix(Q, T), pkg.hamming_pairs(Q, T))
            desc = "knn %dx%d" % (nq, nt)
        log(case, desc, "OK" if ok else "MISMATCH")
        if not ok:
            bad.append(desc)
    return bad


if __name__ == "__main__":
    bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    print("mismatches:", len(bad))
    sys.exit(1 if bad else 0)
