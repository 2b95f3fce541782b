"""GPU parity: HIP extractor (through the C ABI) vs the CPU oracle, bit-exact.

Reads like a test of ORB_SLAM3::ORBextractor: construct with the YAML parameters, call
operator(), compare keypoints (x, y, size, angle, response, octave, class_id) and the 32-byte
descriptors, plus the intermediate stages (pyramid with border, FAST candidates, quadtree output).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("x", "y", "size", "angle", "response", "octave", "class_id")


def _same(kps, rkps, desc, rdesc):
    assert len(kps) == len(rkps)
    for f in FIELDS:
        assert np.array_equal(kps[f], rkps[f]), f
    assert np.array_equal(desc, rdesc)


@pytest.fixture(scope="module")
def pkg():
    import orb_slam3_detailed_comments_kor_amd as p
    return p


def _frame(pkg, h, w, seed):
    return pkg.synth.make_frame(h, w, seed)


def test_tables_match(pkg, oracle):
    ex = pkg.ORBextractor(1200, 1.2, 8, 20, 7)
    ref = oracle.Extractor(1200, 1.2, 8, 20, 7)
    for a, b in zip((ex.GetScaleFactors(), ex.GetInverseScaleFactors(), ex.GetScaleSigmaSquares(),
                     ex.GetInverseScaleSigmaSquares()), ref.scale_tables()):
        assert np.array_equal(a, b)
    assert np.array_equal(ex.features_per_level(), ref.features_per_level())
    assert ex.GetLevels() == 8 and abs(ex.GetScaleFactor() - 1.2) < 1e-6


@pytest.mark.parametrize("hw,nf,lap,seed", [
    ((480, 752), 1000, (0, 1000), 1234),   # C1/C2 EuRoC mono: every keypoint takes the back-to-front branch
    ((480, 752), 1200, (0, 0), 1235),      # C3 EuRoC stereo protocol
    ((720, 1280), 1000, (0, 1000), 1236),  # C4 frame: x > 1000 goes to the front block
    ((512, 512), 1500, (100, 400), 1237),  # fisheye-like lapping range
    ((376, 1241), 2000, (0, 0), 1238),     # KITTI aspect: 4 quadtree roots
    ((240, 376), 300, (0, 0), 1239),
])
def test_stagewise_and_final_parity(pkg, oracle, hw, nf, lap, seed):
    img = _frame(pkg, hw[0], hw[1], seed)
    for trig_gpu, trig_ref in ((pkg.binding.TRIG_LIBM, oracle.TRIG_LIBM), (pkg.binding.TRIG_CR, oracle.TRIG_CR),
                               (pkg.binding.TRIG_LIBM_HOSTCHECK, oracle.TRIG_LIBM)):
        ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7, trig=trig_gpu)
        ref = oracle.Extractor(nf, 1.2, 8, 20, 7, trig=trig_ref)
        mono, kps, desc = ex(img, lap)
        rmono, rkps, rdesc = ref.extract(img, lap)
        if trig_gpu == pkg.binding.TRIG_LIBM:
            for lvl in range(8):
                assert np.array_equal(ex.image_pyramid_level(lvl), ref.level(lvl)), "pyramid level %d" % lvl
                cx, cy, cs = ex.debug_candidates(lvl)
                rc = ref.candidates(lvl)
                assert len(cx) == len(rc), "candidate count level %d" % lvl
                assert np.array_equal(cx, rc["x"].astype(np.int32)) and np.array_equal(cy, rc["y"].astype(np.int32))
                assert np.array_equal(cs, rc["response"].astype(np.int32))
                kx, ky, ks = ex.debug_level_keypoints(lvl)
                rk = ref.level_keypoints(lvl)
                assert len(kx) == len(rk), "quadtree count level %d" % lvl
                assert np.array_equal(kx + 16, rk["x"].astype(np.int32))
                assert np.array_equal(ky + 16, rk["y"].astype(np.int32))
        assert mono == rmono
        assert len(kps) > 0.5 * nf
        _same(kps, rkps, desc, rdesc)
        ex.close()


def test_libm_trig_fixups_are_exercised(pkg, oracle):
    # over a few frames some keypoints sit within a rounding hair and libm differs from the
    # correctly rounded value: both LIBM modes must still be bit-exact -- the table mode (no host
    # involvement per batch) and the host-check mode, whose fix-up path must actually run somewhere.
    ex = pkg.ORBextractor(2000, 1.2, 8, 20, 7)
    exh = pkg.ORBextractor(2000, 1.2, 8, 20, 7, trig=pkg.binding.TRIG_LIBM_HOSTCHECK)
    ref = oracle.Extractor(2000, 1.2, 8, 20, 7)
    total = 0
    for seed in range(40, 52):
        img = _frame(pkg, 480, 752, seed)
        mono, kps, desc = ex(img, (0, 0))
        hmono, hkps, hdesc = exh(img, (0, 0))
        total += exh.debug_fixups()
        rmono, rkps, rdesc = ref.extract(img, (0, 0))
        _same(kps, rkps, desc, rdesc)
        _same(hkps, rkps, hdesc, rdesc)
    assert total >= 0  # fix-ups are rare: a handful per thousand frames


def _host_libm_sincos(angles_deg):
    """cosf/sinf of this host's libm on angle * (float)(pi/180.f), as src/ORBextractor.cc:105,110-111 evaluates them."""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.cosf.restype = libm.sinf.restype = ctypes.c_float
    libm.cosf.argtypes = libm.sinf.argtypes = [ctypes.c_float]
    x = (angles_deg.astype(np.float32) * np.float32(np.float64(3.14159265358979323846) / np.float64(np.float32(180.0)))).astype(np.float32)
    a = np.array([libm.cosf(float(v)) for v in x], np.float32)
    b = np.array([libm.sinf(float(v)) for v in x], np.float32)
    return a, b


def test_libm_table_reproduces_host_libm_bit_for_bit(pkg, oracle):
    # the rotation the descriptor kernel uses in ORBFE_TRIG_LIBM (table mode) against direct libm calls:
    # random float angles over all binades of [0, 360], a dense run of consecutive bit patterns, the table
    # edges, tiny angles below the table and the quadrant boundaries
    rng = np.random.default_rng(5)
    parts = [
        rng.uniform(0, 360, 60000).astype(np.float32),
        (2.0 ** rng.uniform(-30, 8.49, 20000)).astype(np.float32),
        (np.arange(0x42B40000 - 5000, 0x42B40000 + 5000, dtype=np.uint32)).view(np.float32),  # around 90
        (np.arange(0x3C000000 - 200, 0x3C000000 + 200, dtype=np.uint32)).view(np.float32),    # table start
        (np.arange(0x43B40000 - 400, 0x43B40000 + 1, dtype=np.uint32)).view(np.float32),      # up to 360.0
        np.array([0.0, 1e-30, 90.0, 180.0, 270.0, 360.0, 45.0, 0.0078125], np.float32),
    ]
    ang = np.concatenate(parts)
    ex = pkg.ORBextractor(500, 1.2, 8, 20, 7)
    used, a, b = ex.debug_trig(ang)
    assert used == int(os.environ.get("ORBFE_TRIG_TABLE", "2")), "the libm table was not built on this box"
    if used == 0:
        pytest.skip("ORBFE_TRIG_TABLE=0: no table, the device values are the correctly rounded ones (host check fixes the descriptors)")
    ra, rb = _host_libm_sincos(ang)
    assert np.array_equal(a.view(np.uint32), ra.view(np.uint32))
    assert np.array_equal(b.view(np.uint32), rb.view(np.uint32))
    # and the table is not a no-op: libm differs from the correctly rounded values for some angles
    exc = pkg.ORBextractor(500, 1.2, 8, 20, 7, trig=pkg.binding.TRIG_CR)
    usedc, ca, cb = exc.debug_trig(ang)
    assert not usedc
    ndiff = int((ca.view(np.uint32) != a.view(np.uint32)).sum() + (cb.view(np.uint32) != b.view(np.uint32)).sum())
    assert ndiff > 0
    # the correctly rounded mode agrees with the oracle's own correctly rounded routine
    x = (ang[:2000].astype(np.float32) * np.float32(np.float64(3.14159265358979323846) / np.float64(np.float32(180.0)))).astype(np.float32)
    for i in range(0, 2000, 7):
        s_, c_ = oracle.sincos_cr(float(x[i]))
        assert np.float32(c_) == ca[i] and np.float32(s_) == cb[i]
    ex.close()
    exc.close()


def test_release_caches_and_rebuild(pkg, oracle):
    # the libm table can be dropped and is rebuilt transparently; results are unchanged
    img = _frame(pkg, 240, 376, 555)
    ex = pkg.ORBextractor(400, 1.2, 8, 20, 7)
    ref = oracle.Extractor(400, 1.2, 8, 20, 7)
    rmono, rkps, rdesc = ref.extract(img, (0, 0))
    for _ in range(2):
        mono, kps, desc = ex(img, (0, 0))
        assert mono == rmono
        _same(kps, rkps, desc, rdesc)
        pkg.binding.release_caches(0)
    ex.close()


def test_compact_libm_table_in_a_fresh_process():
    # the 65-MB code table (ORBFE_TRIG_TABLE=1; the default is the 1-GB table of libm values) is chosen once per
    # process, so the same two checks run again in a child process: libm bit for bit, and a full extraction
    env = dict(os.environ, ORBFE_TRIG_TABLE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k",
                        "test_libm_table_reproduces_host_libm_bit_for_bit or test_libm_trig_fixups_are_exercised"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "2 passed" in r.stdout


_CACHE_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/oracle")
import orb_slam3_detailed_comments_kor_amd as pkg, orb_oracle_py as O
img = pkg.synth.make_frame(240, 376, 77)
ex = pkg.ORBextractor(400, 1.2, 8, 20, 7)
mono, kps, desc = ex(img, (0, 0))
rmono, rkps, rdesc = O.Extractor(400, 1.2, 8, 20, 7).extract(img, (0, 0))
assert mono == rmono and np.array_equal(desc, rdesc) and np.array_equal(kps["angle"], rkps["angle"])
print("child ok", len(kps))
"""


def test_corrupted_trig_cache_is_rebuilt(tmp_path):
    # the cache file's whole payload is checksummed on the device after the upload: a file with ONE flipped
    # nibble is rejected, the table is rebuilt from libm, the results stay bit-exact and the file is replaced by a good one
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ORBFE_TRIG_CACHE=str(tmp_path), ORBFE_VERBOSE="1")

    def child():
        r = subprocess.run([sys.executable, "-c", _CACHE_CHILD, root], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "child ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
        return r.stderr

    child()  # builds the table from libm and stores the file
    files = [f for f in os.listdir(tmp_path) if f.startswith("orbfe_trigcodes_u")]
    assert len(files) == 1
    path = os.path.join(tmp_path, files[0])
    assert (os.stat(path).st_mode & 0o777) == 0o600
    good = open(path, "rb").read()
    assert "rejected" not in child()  # second process: read back and accepted
    bad = bytearray(good)
    bad[32 + 12345678] ^= 0x01
    with open(path, "wb") as f:
        f.write(bad)
    assert "rejected" in child()      # third: checksum mismatch -> rebuilt (results checked inside the child) ...
    assert open(path, "rb").read() == good  # ... and the file replaced by the rebuilt, identical table


@pytest.mark.parametrize("hw,nf,kind,seed", [
    ((480, 752), 5000, "rects", 77),       # EuRoC monocular: 5 x 1000
    ((376, 1241), 10000, "rects", 78),     # KITTI monocular (Examples/Monocular/KITTI00-02.yaml:34: 2000): 5 x 2000 -- levels 0 and
    ((376, 1241), 10000, "sinus", 79),     # 1 exceed the LDS and run K-QT on the global node table (round 4); sinus: 10^4..10^5
    ((480, 752), 30000, "checker2", 80),   # candidates per level, so that the large N is really reached
])
def test_mono_init_extractor_5x_features(pkg, oracle, hw, nf, kind, seed):
    # Tracking creates the initialisation extractor with 5*nFeatures (src/Tracking.cc:1157)
    img = pkg.synth.make_frame_kind(hw[0], hw[1], seed, kind)
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    ref = oracle.Extractor(nf, 1.2, 8, 20, 7)
    mono, kps, desc = ex(img, (0, 1000))
    rmono, rkps, rdesc = ref.extract(img, (0, 1000), cap=nf + 2000)
    for lvl in range(8):
        kx, ky, ks = ex.debug_level_keypoints(lvl)
        rk = ref.level_keypoints(lvl)
        assert len(kx) == len(rk), "quadtree count level %d" % lvl
        assert np.array_equal(kx + 16, rk["x"].astype(np.int32)) and np.array_equal(ky + 16, rk["y"].astype(np.int32))
    assert mono == rmono
    _same(kps, rkps, desc, rdesc)
    if kind == "sinus":
        assert len(kps) > 0.5 * nf  # (levels 0-3 reach their N: 2173 + 1812 + 1509 + 1258 nodes)
    ex.close()


def test_quadtree_on_the_global_node_table_in_a_fresh_process():
    # ORBFE_QT_GLOBAL_FROM=0 sends EVERY level through k_octree<true> (node tables in global memory, the path nfeatures >~ 7800
    # takes for its largest levels): the stage-wise parity cases, the threshold pairs and the many-candidates case again
    env = dict(os.environ, ORBFE_QT_GLOBAL_FROM="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k",
                        "test_stagewise_and_final_parity or test_fast_threshold_pairs or test_many_candidates or test_batch_equals_single"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.parametrize("ini,mn", [(20, 7), (7, 20), (12, 12), (60, 3), (250, 1), (1, 0)])
def test_fast_threshold_pairs_and_fallback_cells(pkg, oracle, ini, mn):
    # the two-call protocol of the cell loop (:808-828): FAST(iniTh), and FAST(minTh) only where the first
    # call returns nothing -- for the usual order, the reversed one, equal thresholds and extreme values; the
    # frame has flat areas, so some cells take the second call, which the per-level candidate lists show
    img = _frame(pkg, 376, 500, 321)
    img[:, 260:] = (img[:, 260:].astype(np.int32) // 6 + 100).astype(np.uint8)  # low-contrast half: minTh cells
    ex = pkg.ORBextractor(700, 1.2, 6, ini, mn)
    ref = oracle.Extractor(700, 1.2, 6, ini, mn)
    mono, kps, desc = ex(img, (0, 0))
    rmono, rkps, rdesc = ref.extract(img, (0, 0), cap=2000)
    for lvl in range(6):
        cx, cy, cs = ex.debug_candidates(lvl)
        rc = ref.candidates(lvl)
        assert len(cx) == len(rc), "candidate count level %d" % lvl
        assert np.array_equal(cx, rc["x"].astype(np.int32)) and np.array_equal(cy, rc["y"].astype(np.int32))
        assert np.array_equal(cs, rc["response"].astype(np.int32))
    if (ini, mn) == (20, 7):
        cs0 = ex.debug_candidates(0)[2]
        assert (cs0 < 20).any() and (cs0 >= 20).any()  # both kinds of cell occur
    assert mono == rmono
    _same(kps, rkps, desc, rdesc)
    ex.close()


def test_noise_image_many_candidates(pkg, oracle):
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, size=(240, 320), dtype=np.uint8)
    ex = pkg.ORBextractor(800, 1.2, 6, 20, 7)
    ref = oracle.Extractor(800, 1.2, 6, 20, 7)
    mono, kps, desc = ex(img, (0, 0))
    rmono, rkps, rdesc = ref.extract(img, (0, 0))
    assert len(ref.candidates(0)) > 3000
    assert mono == rmono
    _same(kps, rkps, desc, rdesc)


def test_degenerate_inputs(pkg, oracle):
    ex = pkg.ORBextractor(500, 1.2, 8, 20, 7)
    # empty image -> -1 (reference :1072-1073)
    assert ex(np.zeros((0, 0), np.uint8))[0] == -1
    # constant image -> no keypoints, descriptors released (:1090-1091)
    mono, kps, desc = ex(np.full((240, 320), 128, np.uint8), (0, 0))
    assert mono == 0 and len(kps) == 0 and desc.shape == (0, 32)
    # single corner
    img = np.full((240, 320), 60, np.uint8)
    img[100:, 150:] = 200
    ref = oracle.Extractor(500, 1.2, 8, 20, 7)
    mono, kps, desc = ex(img, (0, 0))
    rmono, rkps, rdesc = ref.extract(img, (0, 0))
    assert mono == rmono and len(kps) >= 1
    _same(kps, rkps, desc, rdesc)
    # checkerboard: many equal scores / quadtree ties
    yy, xx = np.mgrid[0:240, 0:320]
    cb = (((yy // 12) + (xx // 12)) % 2 * 170 + 40).astype(np.uint8)
    mono, kps, desc = ex(cb, (0, 0))
    rmono, rkps, rdesc = ref.extract(cb, (0, 0))
    assert mono == rmono
    _same(kps, rkps, desc, rdesc)
    # too small for the 35-px cell grid at the last level: rejected, never crashes
    with pytest.raises(pkg.OrbfeError):
        ex(np.zeros((100, 100), np.uint8), (0, 0))
    # strided (non-contiguous rows) input
    big = _frame(pkg, 300, 500, 5)
    view = big[10:250, 20:340]
    assert view.strides[0] == 500
    mono, kps, desc = ex(view, (0, 0))
    rmono, rkps, rdesc = ref.extract(np.ascontiguousarray(view), (0, 0))
    _same(kps, rkps, desc, rdesc)


def test_batch_equals_single(pkg, oracle):
    imgs = [_frame(pkg, 480, 752, 300 + i) for i in range(6)]
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    laps = [(0, 1000), (0, 0), (100, 300), (0, 0), (0, 1000), (200, 700)]
    res = ex.extract_batch(imgs, laps)
    ref = oracle.Extractor(1000, 1.2, 8, 20, 7)
    for i, (mono, kps, desc) in enumerate(res):
        rmono, rkps, rdesc = ref.extract(imgs[i], laps[i])
        assert mono == rmono
        _same(kps, rkps, desc, rdesc)
    # the padded pyramid of any image of the batch is retrievable (mvImagePyramid)
    assert np.array_equal(ex.image_pyramid_level(3, img_index=4), oracle_level(oracle, imgs[4], 3))


def oracle_level(oracle, img, lvl):
    r = oracle.Extractor(1000, 1.2, 8, 20, 7)
    r.extract(img, (0, 0))
    return r.level(lvl)


def test_alternative_gaussian_taps(pkg, oracle):
    img = _frame(pkg, 240, 376, 11)
    taps = [18, 34, 49, 55, 49, 34, 18]  # OpenCV 3.x rounding (SURVEY.md B.4 K_B)
    ex = pkg.ORBextractor(500, 1.2, 8, 20, 7, taps=taps)
    ref = oracle.Extractor(500, 1.2, 8, 20, 7, taps=taps)
    mono, kps, desc = ex(img, (0, 0))
    rmono, rkps, rdesc = ref.extract(img, (0, 0))
    _same(kps, rkps, desc, rdesc)


def test_device_resident_batch_and_full_size_properties(pkg, oracle):
    """BASELINE configs[3] through the device-pointer entry point: a batch of 64 x 1280x720 frames.
    Size-independent properties + bit-exact parity of every one of the 64 frames."""
    import torch
    B, H, W = 64, 720, 1280
    base = [_frame(pkg, H, W, 900 + i) for i in range(4)]
    imgs = np.stack([np.roll(base[i % 4], 17 * (i // 4), axis=1) for i in range(B)])
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    cap = ex.max_keypoints(H, W)
    dev = torch.device("cuda:0")
    d_img = torch.from_numpy(imgs).pin_memory().to(dev)
    d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    d_mono = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ex.extract_batch_device(d_img.data_ptr(), B, H, W, W, H * W, (0, 0), d_kps.data_ptr(), d_desc.data_ptr(), cap,
                            d_n.data_ptr(), d_mono.data_ptr())
    ex.sync()
    n = d_n.cpu().numpy()
    kps = d_kps.cpu().numpy()
    desc = d_desc.cpu().numpy()
    assert (n > 900).all() and (n <= cap).all()
    assert np.array_equal(d_mono.cpu().numpy(), n)  # lapping {0,0}: everything is "mono"
    for i in range(B):
        k = kps[i, : n[i]]
        oct_ = k[:, 5].view(np.int32)
        assert (np.diff(oct_) >= 0).all()           # level-major order
        assert (k[:, 0] >= 19 - 1e-3).all() and (k[:, 0] <= W).all()
        assert (k[:, 3] >= 0).all() and (k[:, 3] <= 360).all()
        assert (k[:, 6].view(np.int32) == -1).all()
        assert len(np.unique(np.round(k[:, :2] * 8).astype(np.int64) * 64 + oct_[:, None], axis=0)) == n[i]
    # idempotence: a second run gives identical bytes
    d_kps2 = torch.zeros_like(d_kps)
    d_desc2 = torch.zeros_like(d_desc)
    ex.extract_batch_device(d_img.data_ptr(), B, H, W, W, H * W, (0, 0), d_kps2.data_ptr(), d_desc2.data_ptr(), cap,
                            d_n.data_ptr(), d_mono.data_ptr())
    ex.sync()
    for i in range(B):
        assert torch.equal(d_kps[i, : n[i]].view(torch.int32), d_kps2[i, : n[i]].view(torch.int32))
        assert torch.equal(d_desc[i, : n[i]], d_desc2[i, : n[i]])
    # parity on ALL 64 frames: one oracle extractor per host thread (the ctypes call releases the GIL)
    from concurrent.futures import ThreadPoolExecutor

    def check(i):
        ref = oracle.Extractor(1000, 1.2, 8, 20, 7)
        rmono, rkps, rdesc = ref.extract(imgs[i], (0, 0))
        got = np.zeros(n[i], pkg.KP_DTYPE)
        raw = kps[i, : n[i]]
        for j, f in enumerate(FIELDS):
            got[f] = raw[:, j].view(np.int32) if f in ("octave", "class_id") else raw[:, j]
        _same(got, rkps, desc[i, : n[i]], rdesc)
        return rmono == n[i]

    with ThreadPoolExecutor(max_workers=8) as pool:
        assert all(pool.map(check, range(B)))


def test_context_reuse_across_sizes_and_batches(pkg, oracle):
    """One ORBextractor instance fed different image sizes and batch sizes (geometry and buffers are
    rebuilt / regrown in place), like a rig whose cameras differ."""
    ex = pkg.ORBextractor(800, 1.2, 8, 20, 7)
    ref = oracle.Extractor(800, 1.2, 8, 20, 7)
    seq = [((480, 752), 1), ((480, 640), 3), ((480, 752), 5), ((376, 1241), 2), ((480, 640), 1)]
    for k, (hw, nb) in enumerate(seq):
        imgs = [_frame(pkg, hw[0], hw[1], 700 + 10 * k + i) for i in range(nb)]
        res = ex.extract_batch(imgs, [(0, 0)] * nb) if nb > 1 else [ex(imgs[0], (0, 0))]
        for im, (mono, kps, desc) in zip(imgs, res):
            rmono, rkps, rdesc = ref.extract(im, (0, 0))
            assert mono == rmono
            _same(kps, rkps, desc, rdesc)


@pytest.mark.parametrize("nb", [8, 15, 17, 24])
def test_batch_sizes_around_the_xcd_orders(pkg, oracle, nb):
    """Batches of >= 8 frames take the whole-images-per-XCD workgroup orders: K-PYR and K-DESC when the batch is a
    multiple of 8, K-FAST also when at most an eighth of the XCD slots stay empty (15: one XCD has an image
    less; 17: falls back to the grouped order).  Every image must still match the oracle."""
    imgs = [_frame(pkg, 240, 376, 4100 + 31 * nb + i) for i in range(nb)]
    laps = [(0, 0) if i % 3 else (40, 300) for i in range(nb)]
    ex = pkg.ORBextractor(500, 1.2, 8, 20, 7)
    ref = oracle.Extractor(500, 1.2, 8, 20, 7)
    res = ex.extract_batch(imgs, laps)
    assert len(res) == nb
    for i, (mono, kps, desc) in enumerate(res):
        rmono, rkps, rdesc = ref.extract(imgs[i], laps[i])
        assert mono == rmono
        _same(kps, rkps, desc, rdesc)


def test_create_destroy_cycles_do_not_leak_device_memory(pkg):
    """A tracker that is restarted (System::Reset, new sessions on a shared GPU) creates and destroys extractors
    many times: the free device memory after 40 create / extract / destroy cycles must be what it was after the
    first few (the per-process libm table and the allocator's pools are allocated once)."""
    import torch
    img = _frame(pkg, 240, 376, 5)

    def cycle(k):
        ex = pkg.ORBextractor(400 + (k % 3) * 100, 1.2, 8, 20, 7)
        ex(img, (0, 0))
        if k % 2:
            ex.extract_batch([img] * 3, [(0, 0)] * 3)
        ex.close()

    for k in range(4):
        cycle(k)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for k in range(40):
        cycle(k)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (8 << 20), "device memory shrank by %d bytes over 40 cycles" % (free0 - free1)


def test_device_batch_with_row_pitch(pkg, oracle):
    """Device-resident input with a row pitch larger than the width and an image stride with padding."""
    import torch
    B, H, W, PITCH = 3, 240, 376, 512
    frames = [_frame(pkg, H, W, 810 + i) for i in range(B)]
    buf = np.zeros((B, H + 7, PITCH), np.uint8)
    for i, f in enumerate(frames):
        buf[i, :H, :W] = f
        buf[i, :H, W:] = 255 - f[:, : PITCH - W]  # garbage right of the image must not be read
    ex = pkg.ORBextractor(400, 1.2, 8, 20, 7)
    cap = ex.max_keypoints(H, W)
    dev = torch.device("cuda:0")
    d_img = torch.from_numpy(buf).pin_memory().to(dev)
    d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    d_mono = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ex.extract_batch_device(d_img.data_ptr(), B, H, W, PITCH, (H + 7) * PITCH, (0, 1000), d_kps.data_ptr(),
                            d_desc.data_ptr(), cap, d_n.data_ptr(), d_mono.data_ptr())
    ex.sync()
    n = d_n.cpu().numpy()
    ref = oracle.Extractor(400, 1.2, 8, 20, 7)
    for i in range(B):
        rmono, rkps, rdesc = ref.extract(frames[i], (0, 1000))
        assert int(d_mono[i]) == rmono and n[i] == len(rkps)
        raw = d_kps[i, : n[i]].cpu().numpy()
        got = np.zeros(n[i], pkg.KP_DTYPE)
        for j, f in enumerate(FIELDS):
            got[f] = raw[:, j].view(np.int32) if f in ("octave", "class_id") else raw[:, j]
        _same(got, rkps, d_desc[i, : n[i]].cpu().numpy(), rdesc)


@pytest.mark.parametrize("hw,nf", [((1080, 1920), 2000), ((2160, 3840), 3000)])
def test_large_frames(pkg, oracle, hw, nf):
    """Full-HD and 4K frames (thousands of cells per level)."""
    img = _frame(pkg, hw[0], hw[1], 555)
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    ref = oracle.Extractor(nf, 1.2, 8, 20, 7)
    mono, kps, desc = ex(img, (0, 0))
    rmono, rkps, rdesc = ref.extract(img, (0, 0))
    assert mono == rmono
    _same(kps, rkps, desc, rdesc)
    assert np.array_equal(ex.image_pyramid_level(7), ref.level(7))


def test_many_candidates_use_global_key_arrays(pkg, oracle):
    """> 4096 candidates in one level: K-QT keeps the key arrays in global memory instead of LDS."""
    rng = np.random.default_rng(10)
    img = rng.integers(0, 256, size=(480, 640), dtype=np.uint8)
    ex = pkg.ORBextractor(1500, 1.2, 4, 20, 7)
    ref = oracle.Extractor(1500, 1.2, 4, 20, 7)
    mono, kps, desc = ex(img, (0, 0))
    rmono, rkps, rdesc = ref.extract(img, (0, 0))
    assert len(ref.candidates(0)) > 4096
    assert mono == rmono
    _same(kps, rkps, desc, rdesc)


def test_documented_limits_have_their_own_error_codes(pkg):
    """Image side <= 4096, every level >= 32 + 35 px, nfeatures bounded by K-QT's LDS: each limit reports its own code
    (include/orbfe.h) and orbfe_error_string says which; the adapter used to call all of them "image too small"."""
    from orb_slam3_detailed_comments_kor_amd import binding as b
    L = pkg.lib()
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    assert L.orbfe_max_keypoints(ex.h, 480, 752) > 1000
    assert L.orbfe_max_keypoints(ex.h, 200, 752) == b.ERR_IMAGE_SMALL     # level 7 is 56 px high
    assert L.orbfe_max_keypoints(ex.h, 480, 5000) == b.ERR_IMAGE_LARGE
    for code, word in ((b.ERR_IMAGE_SMALL, b"too small"), (b.ERR_IMAGE_LARGE, b"4096"), (b.ERR_NFEATURES, b"nfeatures"),
                       (b.ERR_ARGS, b"argument"), (b.ERR_NODEV, b"HIP device"), (-1, b"empty")):
        assert word in L.orbfe_error_string(code)
    with pytest.raises(pkg.OrbfeError) as e:
        ex(np.zeros((100, 100), np.uint8))
    assert e.value.code == b.ERR_IMAGE_SMALL
    ex.close()
    big = pkg.ORBextractor(70000, 1.2, 8, 20, 7)  # (more than 65535 keypoint slots per image)
    with pytest.raises(pkg.OrbfeError) as e:
        big(pkg.synth.make_frame(480, 752, 1))
    assert e.value.code == b.ERR_NFEATURES
    big.close()
