"""Where the time of BASELINE configs[4]'s extraction call goes: two 1024 x 1024 images, nFeatures 1500, KB8 rays
(orbfe_set_kb8), one blocking orbfe_extract_batch call per stereo frame -- wall time per call with pageable and page-locked
caller images, and the per-stage device times (hipEvents on every call: the absolute call time then reads high, the split is
what matters).  usage: python tools/c5_stages.py"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import orb_slam3_detailed_comments_kor_amd as pkg
H = W = 1024
left, right = pkg.synth.make_stereo_pair(H, W, 51, shift=40)
sc = W / 512.0
P = np.array([190.978477 * sc, 190.973307 * sc, 254.931706 * sc, 256.897442 * sc, 0.003482389, 0.000715034, -0.002053236, 0.000202937], np.float32)
for pinned in (False, True):
    ex = pkg.ORBextractor(1500, 1.2, 8, 20, 7)
    ex.set_kb8(P)
    if pinned:
        buf = pkg.binding.PinnedBuffer(2 * H * W)
        arr = buf.array((2, H, W), np.uint8)
        arr[0], arr[1] = left, right
        imgs = [arr[0], arr[1]]
    else:
        imgs = [left, right]
    laps = [(0, W - 1), (0, W - 1)]
    for _ in range(30):
        ex.extract_batch(imgs, laps)
    t0 = time.perf_counter()
    N = 300
    for _ in range(N):
        res = ex.extract_batch(imgs, laps)
    dt = (time.perf_counter() - t0) / N
    ex.profile(True)
    for _ in range(100):
        ex.extract_batch(imgs, laps)
    st = ex.stage_ms()
    ex.profile(False)
    print("%s images: %.4f ms per call (python binding, result copies included), keypoints %d + %d; device stages (us): %s; sum %.1f us"
          % ("page-locked" if pinned else "pageable", dt * 1e3, len(res[0][1]), len(res[1][1]), {a: round(b * 1e3, 1) for a, b in st.items()},
             1e3 * sum(st.values())))
    ex.close()
