"""Stamps of k_bfknn2_frames_mfma (a -DORBFE_KNN2_TIMING library: tools/ab_build.sh knntime "-DORBFE_KNN2_TIMING -mllvm -amdgpu-mfma-vgpr-form";
ORBFE_LIB=.../liborbfe_knntime.so python tools/knn_times.py): mean over the workgroups, s_memtime ticks (100 MHz) since a
workgroup's start."""
import ctypes as C, numpy as np, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import orb_slam3_detailed_comments_kor_amd as pkg
L = pkg.lib()
dev = torch.device("cuda:0")
njobs, cap = 64, 1008
rng = np.random.default_rng(5)
desc = torch.from_numpy(rng.integers(0, 256, size=(njobs + 1, cap, 32), dtype=np.uint8)).to(dev)
cnt = torch.from_numpy(rng.integers(950, cap + 1, size=njobs + 1).astype(np.int32)).to(dev)
jobs = np.zeros((njobs, 4), np.uint64)
for j in range(njobs):
    jobs[j] = (desc[j].data_ptr(), cnt[j:].data_ptr(), desc[j + 1].data_ptr(), cnt[j + 1:].data_ptr())
d_jobs = torch.from_numpy(jobs.view(np.int64)).to(dev)
d_idx = torch.zeros((njobs, cap, 2), dtype=torch.int32, device=dev)
d_dist = torch.zeros_like(d_idx)
for _ in range(5):
    pkg.binding.bfknn2_frames_device(d_jobs.data_ptr(), njobs, cap, d_idx.data_ptr(), d_dist.data_ptr())
pkg.binding.matcher_sync()
if hasattr(L, "orbfe_debug_knn_times"):
    t = np.zeros(16, np.uint64)
    L.orbfe_debug_knn_times.argtypes = [C.c_void_p]
    L.orbfe_debug_knn_times(t.ctypes.data_as(C.c_void_p))
    for _ in range(10):
        pkg.binding.bfknn2_frames_device(d_jobs.data_ptr(), njobs, cap, d_idx.data_ptr(), d_dist.data_ptr())
    L.orbfe_debug_knn_times(t.ctypes.data_as(C.c_void_p))
    n = max(int(t[7]), 1)
    clk = float(t[5]) / max(float(t[11]), 1.0) * 100.0  # MHz: shader cycles per 100-MHz tick
    print("workgroups %d, clock %.0f MHz; mean shader cycles since a workgroup's start: queries expanded %.0f | packed rows in LDS %.0f | tiles 0, 1 "
          "expanded %.0f | first step %.0f | loop done %.0f | end %.0f (= %.2f us)"
          % ((n, clk) + tuple(float(t[k]) / n for k in (1, 2, 8, 3, 4, 5)) + (float(t[5]) / n / clk,)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st = torch.cuda.current_stream()
pkg.binding.matcher_sync()
torch.cuda.synchronize()
e0.record()
for _ in range(50):
    pkg.binding.bfknn2_frames_device(d_jobs.data_ptr(), njobs, cap, d_idx.data_ptr(), d_dist.data_ptr(), stream=st.cuda_stream)
e1.record()
torch.cuda.synchronize()
print("launch %.2f us" % (e0.elapsed_time(e1) * 1e3 / 50))
