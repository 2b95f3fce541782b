"""The Python form of tools/soak/mc_soak.cpp: the two ctypes tests of tests/test_gpu_multicam.py back to back in ONE process, many
times -- a three-lane context under an RCCL handle, closed, then (where the round-5 process died) a list of
fresh numpy arrays through torch's Tensor.cuda(), a two-lane context under the next communicator.  stderr is not captured, so
a fatal message of glibc / the HIP runtime / RCCL / libstdc++ is seen; faulthandler prints the Python stack of every thread.

usage: python3 tools/soak/mc_soak.py seconds [transport: rccl | host]"""
import faulthandler
import os
import sys
import time

os.environ.setdefault("LIBC_FATAL_STDERR_", "1")
faulthandler.enable(all_threads=True)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

import orb_slam3_detailed_comments_kor_amd as pkg  # noqa: E402
from orb_slam3_detailed_comments_kor_amd import binding  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    transport = binding.MC_HOST if len(sys.argv) > 2 and sys.argv[2] == "host" else binding.MC_RCCL
    rows, cols = 240, 376
    hip = C.CDLL("libamdhip64.so")  # (the HIP runtime torch and liborbfe.so share)
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    base3 = np.stack([pkg.synth.make_frame(rows, cols, 700 + i) for i in range(3)])
    base16 = [np.stack([pkg.synth.make_frame(rows, cols, 1200 + 40 * s + i) for i in range(16)]) for s in range(2)]
    first = {}
    t0 = time.time()
    it = 0
    rng = np.random.default_rng(5)
    while time.time() - t0 < seconds:
        # --- test_ctypes_handle_against_the_oracle without the oracle
        imgs = base3.copy()
        d_img = torch.from_numpy(imgs).cuda()
        ex = pkg.ORBextractor(500, 1.2, 8, 20, 7, device=0)
        ex.set_lanes(3)
        ex.set_lane_input_guard(False)
        cap = ex.max_keypoints(rows, cols)
        mc = binding.MultiCam(ex, None, 0, 1, 3, cap, transport)
        for _ in range(2):
            mc.submit(d_img.data_ptr(), rows, cols, cols, rows * cols, (0, 0))
        mc.wait()
        v = mc.wait()
        idx, dist = mc.match_ring((1, 2))
        g = np.empty(mc.slab_bytes, np.uint8)
        torch.cuda.synchronize()
        assert hip.hipMemcpy(g.ctypes.data_as(C.c_void_p), C.c_void_p(v.gathered), g.size, 2) == 0
        counts = g[mc.count_off:mc.count_off + 12].view(np.int32).copy()
        key = (counts.tobytes(), idx[:, :int(counts.min())].tobytes(), dist[:, :int(counts.min())].tobytes())
        if "a" not in first:
            first["a"] = key
            assert counts.min() > 50
        assert key == first["a"], "iteration %d: results changed" % it
        mc.close()
        ex.close()
        del d_img, idx, dist, g
        # --- the numpy work between the two tests (a second in the round-5 process; here 0 .. 30 ms, sometimes the second)
        r = int(rng.integers(0, 64))
        if r == 0:
            time.sleep(1.0)
        elif r < 16:
            time.sleep(0.002 * r)
        # --- test_two_lane_context_behind_the_exchange: the line the process died in, then the rest
        sets = [a.copy() for a in base16]
        d_sets = [torch.from_numpy(a).cuda() for a in sets]
        ex = pkg.ORBextractor(400, 1.2, 8, 20, 7, device=0)
        ex.set_lanes(2)
        cap = ex.max_keypoints(rows, cols)
        mc = binding.MultiCam(ex, None, 0, 1, 16, cap, transport)
        inflight = 0
        sums = []
        for b in range(6):
            if inflight == binding.MC_MAX_IN_FLIGHT:
                v = mc.wait()
                inflight -= 1
                g = np.empty(mc.slab_bytes, np.uint8)
                assert hip.hipMemcpy(g.ctypes.data_as(C.c_void_p), C.c_void_p(v.gathered), g.size, 2) == 0
                sums.append((int(v.batch) % 2, g[mc.count_off:mc.count_off + 64].tobytes()))
            mc.submit(d_sets[b % 2].data_ptr(), rows, cols, cols, rows * cols, (0, 0))
            inflight += 1
        while inflight:
            mc.wait()
            inflight -= 1
        if "b" not in first:
            first["b"] = dict(sums)
        for s, c in sums:
            assert first["b"][s] == c, "iteration %d: counts changed" % it
        mc.close()
        ex.close()
        del d_sets
        it += 1
        if it % 25 == 0:
            print("%d iterations, %.0f s" % (it, time.time() - t0), flush=True)
    print("mc_soak.py: %d clean iterations (%s) in %.0f s" % (it, "host" if transport == binding.MC_HOST else "rccl", time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
