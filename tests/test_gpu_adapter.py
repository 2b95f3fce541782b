"""The C++ adapter (adapters/ORBextractor.h) keeps the reference's signatures: a C++ caller written like
Frame::ExtractORB gets the same bytes as the oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fnv(desc, kps):
    h = 1469598103934665603
    for b in desc.tobytes() + kps.tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_cpp_adapter_matches_oracle(tmp_path, oracle):
    import orb_slam3_detailed_comments_kor_amd as pkg
    exe = str(tmp_path / "test_adapter")
    libdir = os.path.join(ROOT, "orb_slam3_detailed_comments_kor_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "adapters"),
                           os.path.join(ROOT, "adapters", "test_adapter.cpp"), "-o", exe, "-L" + libdir, "-lorbfe",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    img = pkg.synth.make_frame(480, 752, 4242)
    raw = tmp_path / "img.raw"
    raw.write_bytes(img.tobytes())
    out = subprocess.check_output([exe, str(raw), "480", "752", "1000"], text=True).split("\n")
    mono, n, h, levels, prow, pcol = [int(v) for v in out[0].split()]
    ref = oracle.Extractor(1000, 1.2, 8, 20, 7)
    rmono, rkps, rdesc = ref.extract(img, (0, 1000))
    assert (mono, n, levels) == (rmono, len(rkps), 8)
    assert h == _fnv(rdesc, rkps)
    assert (prow, pcol) == (278, 435)          # mvImagePyramid[3] of a 752x480 frame
    # Frame::ComputeStereoMatches' reads of mvImagePyramid (11x11 windows at the keypoints' octaves), no flag set
    nrows, hp = [int(v) for v in out[1].split()]
    inv = ref.scale_tables()[1]
    h2 = 1469598103934665603
    levels_ = [ref.level(l)[19:-19, 19:-19].astype(np.int64) for l in range(8)]
    for k in rkps:
        sf = np.float32(inv[int(k["octave"])])
        u = int(np.float32(np.float32(k["x"]) * sf) + np.float32(0.5))
        v = int(np.float32(np.float32(k["y"]) * sf) + np.float32(0.5))
        win = levels_[int(k["octave"])][v - 5:v + 6, u - 5:u + 6]
        assert win.shape == (11, 11)
        h2 = ((h2 ^ int(win.sum())) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert nrows == 480 and hp == h2
    assert int(out[2]) == -1                   # empty image -> -1 (:1072-1073)


def test_cpp_adapter_stereo_pair_in_one_call(tmp_path, oracle):
    """ORBextractor::ExtractStereoPair of the adapter: both images in one batched call, ComputeStereoMatches on the
    resident results -- keypoints, descriptors, mvuRight and mvDepth as the oracle's two extractions + matching give them."""
    import orb_slam3_detailed_comments_kor_amd as pkg
    exe = str(tmp_path / "test_adapter")
    libdir = os.path.join(ROOT, "orb_slam3_detailed_comments_kor_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "adapters"),
                           os.path.join(ROOT, "adapters", "test_adapter.cpp"), "-o", exe, "-L" + libdir, "-lorbfe",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    left, right = pkg.synth.make_stereo_pair(480, 752, 77, shift=18)
    (tmp_path / "l.raw").write_bytes(left.tobytes())
    (tmp_path / "r.raw").write_bytes(right.tobytes())
    out = subprocess.check_output([exe, str(tmp_path / "l.raw"), "480", "752", "1200", str(tmp_path / "r.raw")], text=True).split("\n")
    m, nl, nr, hl, hr, hu, ml, mr = [int(v) for v in out[3].split()]
    oL, oR = oracle.Extractor(1200, 1.2, 8, 20, 7), oracle.Extractor(1200, 1.2, 8, 20, 7)
    rml, kL, dL = oL.extract(left, (0, 0))
    rmr, kR, dR = oR.extract(right, (0, 0))
    mbf = np.float32(47.90639384423901)
    mb = np.float32(mbf / np.float32(435.2046959714599))
    rn, ruR, rdep = oracle.compute_stereo_matches(oL, oR, kL, dL, kR, dR, float(mb), float(mbf))
    assert (nl, nr, ml, mr) == (len(kL), len(kR), rml, rmr)
    assert hl == _fnv(dL, kL) and hr == _fnv(dR, kR)
    assert m == rn and m > 100
    assert hu == _fnv(ruR.astype(np.float32), rdep.astype(np.float32))
