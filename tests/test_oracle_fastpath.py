"""The oracle's TIMING-ONLY fast path (SIMD prefilter in FAST, vector-friendly blur; oracle/orb_oracle.cpp) gives exactly what the
scalar parity path gives -- keypoints, descriptors, candidates per level, blurred levels -- on synthetic frames, photographs
and odd sizes.  It exists so that bench.py's cpu_baseline can time a CPU path that is computed the way OpenCV computes it
(VERDICT r05 #9); no parity test uses it as its checker."""
import numpy as np
import pytest

from natural import frame as natural_frame


def _frames():
    import orb_slam3_detailed_comments_kor_amd as pkg
    out = [pkg.synth.make_frame(480, 752, 11), pkg.synth.make_frame(240, 376, 12), pkg.synth.make_frame(271, 347, 13)]
    try:
        out += [natural_frame("china", 480, 752), natural_frame("flower", 427, 640)]
    except Exception:  # noqa: BLE001
        pass
    rng = np.random.default_rng(1)
    out.append(rng.integers(0, 256, (300, 400), dtype=np.uint8))           # noise: every pixel passes the rejection test
    out.append(np.full((260, 300), 255, np.uint8))                         # saturated: v + t clamps
    out.append(np.zeros((260, 300), np.uint8))
    return out


@pytest.mark.parametrize("native", [False, True])
def test_fastpath_equals_scalar_path(oracle, native):
    if native and oracle.lib_native() is None:
        pytest.skip("no native build on this host")
    for k, img in enumerate(_frames()):
        for nf, ini, mn in ((1000, 20, 7), (2000, 40, 3), (500, 5, 1)):
            a = oracle.Extractor(nf, 1.2, 8, ini, mn, native=native)
            b = oracle.Extractor(nf, 1.2, 8, ini, mn, native=native)
            simd = b.set_fastpath(True)
            ra = a.extract(img, (0, 1000))
            rb = b.extract(img, (0, 1000))
            assert ra[0] == rb[0] and np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2], rb[2]), (k, nf, simd)
            for lvl in range(8):
                ca, cb = a.candidates(lvl), b.candidates(lvl)
                assert np.array_equal(ca, cb), (k, nf, lvl)
                if len(a.level_keypoints(lvl)):
                    assert np.array_equal(a.blurred(lvl), b.blurred(lvl)), (k, nf, lvl)


def test_stage_seconds_add_up(oracle):
    import orb_slam3_detailed_comments_kor_amd as pkg
    frames = np.stack([pkg.synth.make_frame(240, 376, 20 + i) for i in range(2)])
    n, t, st = oracle.extract_many_stages(frames, 1, 6, 500, fastpath=False)
    assert n > 0 and set(st) == set(oracle.STAGES) and all(v > 0 for v in st.values())
    assert 0.7 * t < sum(st.values()) <= t * 1.01   # the stages are the call, minus allocation and packing
    n2, t2, st2 = oracle.extract_many_stages(frames, 1, 6, 500, fastpath=True)
    assert n2 == n
