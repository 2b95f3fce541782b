"""Stage times of the resident single-frame / stereo-pair call (events on every call: the absolute call time reads high,
the split is what matters).  usage: [ORBFE_PYR_TILE=n] python tools/single_frame_stages.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench
import orb_slam3_detailed_comments_kor_amd as pkg
H, W = 480, 752
f = bench.bench_frames(H, W, 2)
d = torch.from_numpy(f).cuda()
ex = pkg.ORBextractor(1000)
cap = ex.max_keypoints(H, W)
k = torch.zeros((2, cap, 7), dtype=torch.float32, device='cuda'); de = torch.zeros((2, cap, 32), dtype=torch.uint8, device='cuda')
n = torch.zeros(2, dtype=torch.int32, device='cuda'); m = torch.zeros(2, dtype=torch.int32, device='cuda')
for nimg in (1, 2):
    for it in range(300):
        ex.extract_batch_device(d.data_ptr(), nimg, H, W, W, H * W, (0, 1000), k.data_ptr(), de.data_ptr(), cap, n.data_ptr(), m.data_ptr())
    ex.sync()
    ex.profile(True)
    t0 = time.perf_counter()
    for it in range(2000):
        ex.extract_batch_device(d.data_ptr(), nimg, H, W, W, H * W, (0, 1000), k.data_ptr(), de.data_ptr(), cap, n.data_ptr(), m.data_ptr())
    ex.sync()
    dt = (time.perf_counter() - t0) / 2000
    print("nimg", nimg, "ms/call %.4f" % (dt * 1e3), {a: round(b * 1e3, 1) for a, b in ex.stage_ms().items()})
    ex.profile(False)
