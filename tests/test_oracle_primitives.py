"""Known-answer / closed-form tests that pin the CPU oracle.

The reference has no tests of its own ("parity unpinned"); these are first-principles
checks of every primitive the oracle restates."""
import hashlib
import math
import os
import re
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ----------------------------------------------------------------- ctor tables
def test_scale_tables(oracle):
    e = oracle.Extractor(1000, 1.2, 8)
    sf, inv, s2, is2 = e.scale_tables()
    exp = [1, 1.2000000477, 1.4400000572, 1.7280001640, 2.0736002922, 2.4883203506, 2.9859845638, 3.5831816196]
    assert np.allclose(sf, exp, rtol=0, atol=2e-7)
    # float(float * double(1.2f)) recurrence, reference src/ORBextractor.cc:418
    v = np.float32(1.0)
    for i in range(1, 8):
        v = np.float32(np.float64(v) * np.float64(np.float32(1.2)))
        assert sf[i] == v
        assert s2[i] == np.float32(v * v)
        assert inv[i] == np.float32(1.0) / v
        assert is2[i] == np.float32(1.0) / np.float32(v * v)


@pytest.mark.parametrize("nf,exp", [
    (1000, [217, 181, 151, 126, 105, 87, 73, 60]),
    (1200, [261, 217, 181, 151, 126, 105, 87, 72]),
    (1500, [326, 271, 226, 189, 157, 131, 109, 91]),
    (2000, [434, 362, 302, 251, 209, 175, 145, 122]),
    (5000, [1086, 905, 754, 628, 524, 436, 364, 303]),
])
def test_features_per_level(oracle, nf, exp):
    assert oracle.Extractor(nf).features_per_level().tolist() == exp


def test_umax(oracle):
    u = oracle.Extractor().umax().tolist()
    assert u == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert sum(2 * x + 1 for x in u[1:]) * 2 + 31 == 749  # circular patch area


def test_pattern_tables_identical_and_hashed():
    def load(path):
        txt = open(path).read().split("/*PATTERN-BEGIN*/", 1)[1]
        return [int(x) for x in re.findall(r"-?\d+", txt)]

    a = load(os.path.join(ROOT, "oracle", "orb_pattern.inc"))
    b = load(os.path.join(ROOT, "orb_slam3_detailed_comments_kor_amd", "csrc", "orb_pattern.inc"))
    assert a == b and len(a) == 1024 and sum(a) == -406 and min(a) == -13 and max(a) == 12
    sha = hashlib.sha256(struct.pack("<1024i", *a)).hexdigest()
    assert sha == "7e645581387b82784797e8adddb9b6f0c12611859fda09ca8a9bec96d767a05f"
    assert max(math.hypot(a[i], a[i + 1]) for i in range(0, 1024, 2)) < 18.5  # rotated tap reach <= 18


@pytest.mark.parametrize("wh,sizes", [
    ((752, 480), [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231), (302, 193), (252, 161), (210, 134)]),
    ((1280, 720), [(1280, 720), (1067, 600), (889, 500), (741, 417), (617, 347), (514, 289), (429, 241), (357, 201)]),
])
def test_pyramid_sizes(oracle, wh, sizes):
    e = oracle.Extractor(100)
    img = np.full((wh[1], wh[0]), 90, np.uint8)
    r, k, d = e.extract(img, (0, 0))
    assert r == 0 and len(k) == 0  # constant image: no keypoints (reference :1090-1091)
    for lvl, (w, h) in enumerate(sizes):
        L = e.level(lvl)
        assert L.shape == (h + 38, w + 38)
        assert (L == 90).all()


# ------------------------------------------------------------------- resize
def _resize_np(src, dh, dw):
    """Independent vectorised restatement of SURVEY.md B.1."""
    sh, sw = src.shape
    sx_scale = 1.0 / (float(dw) / sw)
    sy_scale = 1.0 / (float(dh) / sh)
    fx = ((np.arange(dw) + 0.5) * sx_scale - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int64)
    fx = fx - sx.astype(np.float32)
    lo = sx < 0
    fx[lo], sx[lo] = 0, 0
    hi = sx >= sw - 1
    fx[hi], sx[hi] = 0, sw - 1
    a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64)
    a1 = np.rint(fx * np.float32(2048)).astype(np.int64)
    fy = ((np.arange(dh) + 0.5) * sy_scale - 0.5).astype(np.float32)
    sy = np.floor(fy).astype(np.int64)
    fy = fy - sy.astype(np.float32)
    b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int64)
    b1 = np.rint(fy * np.float32(2048)).astype(np.int64)
    y0 = np.clip(sy, 0, sh - 1)
    y1 = np.clip(sy + 1, 0, sh - 1)
    x1 = np.minimum(sx + 1, sw - 1)
    S = src.astype(np.int64)
    H0 = S[y0][:, sx] * a0 + S[y0][:, x1] * a1
    H1 = S[y1][:, sx] * a0 + S[y1][:, x1] * a1
    v = (((b0[:, None] * (H0 >> 4)) >> 16) + ((b1[:, None] * (H1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def test_resize_linear(oracle):
    rng = np.random.default_rng(0)
    src = rng.integers(0, 256, size=(97, 131), dtype=np.uint8)
    assert np.array_equal(oracle.resize_linear(src, 97, 131), src)  # same size: fx = fy = 0
    assert (oracle.resize_linear(np.full((50, 60), 77, np.uint8), 42, 50) == 77).all()
    for dh, dw in [(81, 109), (400, 627), (30, 200)]:
        assert np.array_equal(oracle.resize_linear(src, dh, dw), _resize_np(src, dh, dw))
    # a horizontal ramp stays (approximately) a ramp; exact bilinear within 1 grey level
    ramp = np.tile(np.arange(0, 240, 2, dtype=np.uint8), (20, 1))
    out = oracle.resize_linear(ramp, 20, 100)
    xs = (np.arange(100) + 0.5) * 1.2 - 0.5
    ideal = np.clip(2 * xs, 0, 238)
    assert np.abs(out[5].astype(float) - ideal).max() <= 1.0


def test_border_reflect101(oracle):
    a = np.arange(20, dtype=np.uint8).reshape(4, 5) * 3
    b = oracle.border_reflect101(a, 3)
    assert b.shape == (10, 11)
    assert np.array_equal(b[3:7, 3:8], a)
    assert np.array_equal(b, np.pad(a, 3, mode="reflect"))  # numpy 'reflect' == REFLECT_101


# --------------------------------------------------------------------- FAST
RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
        (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def _is_corner(p, t):
    v = int(p[3, 3])
    r = [int(p[3 + dy, 3 + dx]) for dx, dy in RING]
    for sign in (1, -1):
        ok = [(x - v) * sign > t for x in r]
        ok2 = ok + ok
        run = 0
        for o in ok2:
            run = run + 1 if o else 0
            if run >= 9:
                return True
    return False


def test_fast_score_is_largest_threshold(oracle):
    rng = np.random.default_rng(1)
    checked = 0
    for _ in range(3000):
        p = rng.integers(0, 256, size=(7, 7), dtype=np.uint8)
        if rng.uniform() < 0.7:  # make corners likely: a bright/dark arc
            start = rng.integers(0, 16)
            ln = rng.integers(9, 14)
            base = int(rng.integers(40, 200))
            delta = int(rng.integers(8, 50)) * (1 if rng.uniform() < 0.5 else -1)
            p[3, 3] = base
            for k in range(16):
                dx, dy = RING[k]
                inarc = ((k - start) % 16) < ln
                p[3 + dy, 3 + dx] = np.clip(base + (delta if inarc else 0) + rng.integers(-4, 5), 0, 255)
        s = oracle.fast_score_closed(p)
        for t in (7, 20):
            c = _is_corner(p, t)
            assert c == (s >= t)
            if c:
                assert oracle.fast_score_2loop(p, t) == s  # OpenCV's two-loop cornerScore<16>
        if s >= 0:
            assert _is_corner(p, s) and not _is_corner(p, s + 1)
            checked += 1
    assert checked > 500


def _fast_np(img, t):
    h, w = img.shape
    score = np.zeros((h, w), np.int64)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            p = img[y - 3:y + 4, x - 3:x + 4]
            if _is_corner(p, t):
                # brute-force score
                s = t
                while _is_corner(p, s + 1):
                    s += 1
                score[y, x] = s
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = score[y, x]
            if s == 0 and not _is_corner(img[y - 3:y + 4, x - 3:x + 4], t):
                continue
            nb = score[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if (s > nb).all():
                out.append((x, y, s))
    return out


def test_fast_detect_small_image(oracle):
    from orb_slam3_detailed_comments_kor_amd import synth
    img = synth.make_frame(48, 64, 3, nrect=25)
    for t in (7, 20):
        ref = _fast_np(img, t)
        got = oracle.fast(img, t, True)
        assert [(int(k["x"]), int(k["y"]), int(k["response"])) for k in got] == ref
        assert len(ref) > 0
    assert (got["size"] == 7).all() and (got["angle"] == -1).all()


def test_fast_nms_is_strict(oracle):
    # two adjacent pixels with identical score: neither survives strict '>' NMS
    img = np.full((12, 13), 50, np.uint8)
    img[4:9, 5:8] = 200  # symmetric blob -> symmetric scores
    k = oracle.fast(img, 20, True)
    k_all = oracle.fast(img, 20, False)
    assert len(k_all) >= len(k)
    sc = {(int(a["x"]), int(a["y"])): a["response"] for a in k}
    for (x, y), s in sc.items():
        for (x2, y2), s2 in sc.items():
            if (x, y) != (x2, y2):
                assert max(abs(x - x2), abs(y - y2)) > 1


# --------------------------------------------------------------------- blur
def test_gaussian_blur(oracle):
    assert (oracle.gaussian_blur7(np.full((20, 30), 113, np.uint8)) == 113).all()
    taps = np.array([18, 34, 48, 56, 48, 34, 18])
    assert taps.sum() == 256
    # float kernel (sigma 2, n 7) x 256 and the error-diffusion rounding of SURVEY.md B.4
    g = np.exp(-0.5 * (np.arange(7) - 3) ** 2 / 4.0)
    g = g / g.sum() * 256
    assert np.allclose(g[:4], [17.960788, 33.555169, 48.822483, 55.323121], atol=1e-5)
    err, q = 0.0, []
    for i in range(3):
        v = g[i] + err
        q.append(int(np.rint(v)))
        err = v - q[-1]
    assert q + [256 - 2 * sum(q)] + q[::-1] == taps.tolist()
    imp = np.zeros((21, 21), np.uint8)
    imp[10, 10] = 255
    out = oracle.gaussian_blur7(imp)
    exp = (np.outer(taps, taps) * 255 + 32768) >> 16
    assert np.array_equal(out[7:14, 7:14], exp)
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(33, 41), dtype=np.uint8)
    pad = np.pad(img.astype(np.int64), 3, mode="reflect")
    H = sum(taps[i] * pad[:, i:i + 41] for i in range(7))
    V = sum(taps[j] * H[j:j + 33, :] for j in range(7))
    assert np.array_equal(oracle.gaussian_blur7(img), ((V + 32768) >> 16).astype(np.uint8))
    # alternative tap table (OpenCV 3.x rounding, SURVEY.md B.4 K_B) is a runtime parameter
    out_b = oracle.gaussian_blur7(img, [18, 34, 49, 55, 49, 34, 18])
    assert out_b.shape == img.shape


# ------------------------------------------------------------ atan2 / trig
def test_fast_atan2(oracle):
    assert oracle.fast_atan2(0, 0) == 0
    assert oracle.fast_atan2(0, 5) == 0
    assert oracle.fast_atan2(0, -5) == 180
    assert abs(oracle.fast_atan2(3, 0) - 90) < 1e-4
    assert abs(oracle.fast_atan2(-3, 0) - 270) < 1e-4
    rng = np.random.default_rng(3)
    for _ in range(2000):
        y, x = rng.integers(-90000, 90000, size=2)
        a = oracle.fast_atan2(float(y), float(x))
        ref = math.degrees(math.atan2(y, x)) % 360.0
        d = abs(a - ref)
        assert min(d, 360 - d) < 0.3 and 0 <= a <= 360


def test_sincos_cr_is_correctly_rounded(oracle):
    rng = np.random.default_rng(4)
    factor = np.float32(np.float64(np.pi) / np.float64(np.float32(180.0)))
    bad = 0
    for deg in rng.uniform(0, 360, size=20000).astype(np.float32):
        ang = np.float32(deg * factor)
        s, c = oracle.sincos_cr(ang)
        bad += s != np.float32(np.sin(np.float64(ang)))
        bad += c != np.float32(np.cos(np.float64(ang)))
    assert bad == 0
    assert oracle.sincos_cr(0.0) == (0.0, 1.0)


# ------------------------------------------------------------- IC angle KAT
def test_ic_angle_of_ramps(oracle):
    e = oracle.Extractor(50, 1.2, 1)
    base = np.zeros((80, 100), np.float64)
    yy, xx = np.mgrid[0:80, 0:100]
    for img, expect in [(xx * 2.0, 0.0), (yy * 2.0, 90.0), (200 - xx * 2.0, 180.0), (200 - yy * 2.0, 270.0)]:
        im = np.clip(img, 0, 255).astype(np.uint8)
        im[40, 50] = 255  # something for FAST is not needed: call the moment formula through extract is overkill
        # closed form of the intensity-centroid moments on a linear ramp, checked against fastAtan2
        u = oracle.Extractor().umax()
        m10 = m01 = 0
        for v in range(-15, 16):
            d = u[abs(v)]
            for uu in range(-d, d + 1):
                val = int(im[40 + v, 50 + uu]) if not (v == 0 and uu == 0) else int(np.clip(img, 0, 255)[40, 50])
                m10 += uu * val
                m01 += v * val
        a = oracle.fast_atan2(float(m01), float(m10))
        d = abs(a - expect)
        assert min(d, 360 - d) < 0.5


def test_fast_atan2_fma_variant(oracle):
    """the fused-Horner evaluation (an AVX2 OpenCV build) stays within an ulp or two of the generic one
    and differs from it somewhere -- it is a distinct, selectable arithmetic."""
    rng = np.random.default_rng(4)
    ys, xs = rng.integers(-70000, 70000, 4000), rng.integers(-70000, 70000, 4000)
    a = np.array([oracle.fast_atan2(float(y), float(x)) for y, x in zip(ys, xs)], np.float32)
    b = np.array([oracle.fast_atan2_fma(float(y), float(x)) for y, x in zip(ys, xs)], np.float32)
    assert np.max(np.abs(a - b)) < 1e-4 and (a != b).any()
    ref = (np.degrees(np.arctan2(ys.astype(np.float64), xs.astype(np.float64))) + 360.0) % 360.0
    d = np.abs(b.astype(np.float64) - ref)
    assert np.max(np.minimum(d, 360 - d)) < 0.3
