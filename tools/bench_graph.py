"""Experiment: one extractor step captured in a hipGraph (torch.cuda.CUDAGraph) vs plain launches."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orb_slam3_detailed_comments_kor_amd as pkg
dev = torch.device('cuda', 0)
B, H, W = 64, 480, 752
base = [pkg.synth.make_frame(H, W, 1234 + i) for i in range(8)]
imgs = np.stack([np.roll(base[i % 8], 23 * (i // 8), axis=1) for i in range(B)])
d_img = torch.from_numpy(imgs).to(dev)
ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, device=0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
ex.set_stream(stream.cuda_stream)
cap = ex.max_keypoints(H, W)
d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
d_n = torch.zeros(B, dtype=torch.int32, device=dev)
d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
d_mono = torch.zeros(B, dtype=torch.int32, device=dev)


def step():
    ex.extract_batch_device(d_img.data_ptr(), B, H, W, W, H * W, (0, 1000), d_kps.data_ptr(), d_desc.data_ptr(), cap,
                            d_n.data_ptr(), d_mono.data_ptr())


for _ in range(5):
    step()
torch.cuda.synchronize()
ref_n = d_n.clone()
ref_desc = d_desc.clone()
t0 = time.perf_counter()
for _ in range(50):
    step()
torch.cuda.synchronize()
print("plain   ms/step", 1e3 * (time.perf_counter() - t0) / 50)
g = torch.cuda.CUDAGraph()
d_desc.zero_()
with torch.cuda.graph(g, stream=stream):
    step()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
assert torch.equal(d_n, ref_n) and torch.equal(d_desc, ref_desc), "graph replay differs"
t0 = time.perf_counter()
for _ in range(50):
    g.replay()
torch.cuda.synchronize()
print("graphed ms/step", 1e3 * (time.perf_counter() - t0) / 50)
