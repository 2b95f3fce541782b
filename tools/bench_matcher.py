#!/usr/bin/env python3
"""Matcher micro-benchmark (secondary metrics of SURVEY.md section 8d): per entry point the device time
of the kernel (hipEvents inside the shim), the end-to-end call through the C ABI with host pointers
(H2D + kernel + D2H + host epilogue) and the CPU oracle on one core, on seeded inputs.  One JSON line each."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import orb_slam3_detailed_comments_kor_amd as pkg  # noqa: E402
import orb_oracle_py as O  # noqa: E402
import matcher_inputs as MI  # noqa: E402

pkg.binding.matcher_time_kernels(True)  # kernel_ms below comes from the events inside the shim


def timeit(f, reps):
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    return (time.perf_counter() - t0) / reps


def report(name, units, unit_name, gpu_call, cpu_call, reps=20, algo_bytes=None):
    gpu_call()
    k = []
    for _ in range(reps):
        gpu_call()
        k.append(pkg.binding.matcher_last_kernel_ms())
    kernel_ms = float(np.median(k))
    pkg.binding.matcher_time_kernels(False)  # the call time is measured as a caller sees it: no events, no extra sync
    e2e = timeit(gpu_call, reps)
    pkg.binding.matcher_time_kernels(True)
    cpu = timeit(cpu_call, max(2, reps // 5))
    out = {"op": name, "units": units, "unit": unit_name, "kernel_ms": kernel_ms, "call_ms": 1e3 * e2e,
           "cpu_oracle_ms_1core": 1e3 * cpu, "kernel_units_per_s": units / (kernel_ms * 1e-3),
           "call_units_per_s": units / e2e, "cpu_units_per_s": units / cpu}
    if algo_bytes:
        # SURVEY.md 8(d)'s matcher byte model evaluated: algorithmic bytes / kernel time against the 8 TB/s HBM peak
        out["algorithmic_bytes"] = algo_bytes
        out["kernel_GBps_algorithmic"] = algo_bytes / (kernel_ms * 1e-3) / 1e9
        out["frac_of_hbm_peak"] = out["kernel_GBps_algorithmic"] / 8000.0
    print(json.dumps(out), flush=True)


def main():
    rng = np.random.default_rng(0)
    A = rng.integers(0, 256, size=(1500, 32), dtype=np.uint8)
    B = rng.integers(0, 256, size=(1500, 32), dtype=np.uint8)
    report("hamming_pairs 1500x1500", 1500 * 1500, "distances", lambda: pkg.hamming_pairs(A, B),
           lambda: O.hamming_matrix(A, B), algo_bytes=3000 * 32 + 1500 * 1500 * 2)
    report("bfknn2 1500x1500 (C5 lapping-area brute force)", 1500 * 1500, "distances", lambda: pkg.bfknn2(A, B),
           lambda: O.bfknn2(A, B), algo_bytes=3000 * 32 + 1500 * 16)
    d1, d2, a1, a2 = MI.descriptor_sets(1200, 1200, 5)
    fv1, fv2 = MI.feature_vectors(d1, d2, 5, 10, 2)
    mask1 = (rng.uniform(size=1200) < 0.6).astype(np.uint8)
    npairs = sum(int(fv1[1][i + 1] - fv1[1][i]) * int(fv2[1][j + 1] - fv2[1][j])
                 for i, a in enumerate(fv1[0]) for j, b in enumerate(fv2[0]) if a == b)
    report("SearchByBoW(KF,F) N=1200, 100 nodes", npairs, "candidate pairs",
           lambda: pkg.search_bow(d1, mask1, a1, fv1, d2, None, a2, fv2, 0, 0.7, True),
           lambda: O.search_bow_kf_f(d1, mask1, a1, fv1, d2, a2, fv2, -1, 0.7, True))
    # relocalisation-style batch: 64 (KF, F) problems in one launch
    probs = []
    for k in range(64):
        e1, e2, b1, b2 = MI.descriptor_sets(1200, 1200, 50 + k)
        g1, g2 = MI.feature_vectors(e1, e2, 50 + k, 10, 2)
        probs.append(dict(desc1=e1, mask1=mask1, ang1=b1, fv1=g1, desc2=e2, mask2=None, ang2=b2, fv2=g2, variant=0,
                          nnratio=0.75, check_ori=True))
    npb = 0
    for pr in probs:
        f1, f2 = pr["fv1"], pr["fv2"]
        common = {int(a): i for i, a in enumerate(f1[0])}
        for j, b in enumerate(f2[0]):
            if int(b) in common:
                i = common[int(b)]
                npb += int(f1[1][i + 1] - f1[1][i]) * int(f2[1][j + 1] - f2[1][j])
    # byte model (8d): per shared node the descriptors of both sides + angles, + the match array out
    bow_bytes = sum((len(p["desc1"]) + len(p["desc2"])) * (32 + 4) + len(p["desc2"]) * 4 for p in probs)
    report("SearchByBoW batch of 64 (KF,F) pairs, N=1200", npb, "candidate pairs", lambda: pkg.search_bow_batch(probs),
           lambda: [O.search_bow_kf_f(p["desc1"], p["mask1"], p["ang1"], p["fv1"], p["desc2"], p["ang2"], p["fv2"], -1,
                                      0.75, True) for p in probs], reps=10, algo_bytes=bow_bytes)
    I = MI.tri_inputs(1200, 1200, 9)
    report("SearchForTriangulation_ N=1200", npairs, "candidate pairs",
           lambda: pkg.search_triangulation(I["d1"], I["has1"], I["kp1"], I["a1"], I["oct1"], I["u1"], I["fv1"],
                                            I["d2"], I["has2"], I["kp2"], I["a2"], I["oct2"], I["u2"], I["fv2"],
                                            I["F12"], I["ep"], I["sf"], I["sig"]),
           lambda: O.search_triangulation(I["d1"], I["has1"], I["kp1"], I["a1"], I["oct1"], I["u1"], I["fv1"],
                                          I["d2"], I["has2"], I["kp2"], I["a2"], I["oct2"], I["u2"], I["fv2"],
                                          I["F12"], I["ep"], I["sf"], I["sig"]))
    # Tracking::SearchLocalPoints-sized projection search (1500 local map points into a 2000-feature frame)
    pr = MI.projection_problem(31, n=2000, nq=1500, mode=0, stereo=True, th=1.0, crowd=False)
    report("SearchByProjection(F, local map) N=2000, 1500 points", 1500, "map points",
           lambda: pkg.search_projection(pr), lambda: O.search_projection(pr))
    pr2 = MI.projection_problem(32, n=2000, nq=1500, mode=1, stereo=True, th=15.0, crowd=False, check_orientation=True)
    report("SearchByProjection(Current, Last) N=2000, 1500 points, th=15", 1500, "map points",
           lambda: pkg.search_projection(pr2), lambda: O.search_projection(pr2))
    # the same two searches against a resident frame (orbfe_frame: frame arrays and grid on the device once)
    fr = pkg.ProjectionFrame(pr)
    report("SearchByProjection(F, local map) on a resident frame (orbfe_frame)", 1500, "map points",
           lambda: fr.search(pr), lambda: O.search_projection(pr))
    fr.close()
    fr2 = pkg.ProjectionFrame(pr2)
    report("SearchByProjection(Current, Last) on a resident frame (orbfe_frame)", 1500, "map points",
           lambda: fr2.search(pr2), lambda: O.search_projection(pr2))
    fr2.close()
    prs = [MI.projection_problem(100 + k, n=2000, nq=1500, mode=0, stereo=True, th=1.0, crowd=False) for k in range(64)]
    # byte model: frame side (descriptor 32 + x, y, octave, uRight 16 per feature) + query side (descriptor 32 + x, y, r,
    # levels, xr 24 per map point) in, one match per query and per feature out
    proj_bytes = 64 * (2000 * (32 + 16) + 1500 * (32 + 24) + 1500 * 4 + 2000 * 4)
    report("SearchByProjection batch of 64 (F, local map) N=2000, 1500 points", 64 * 1500, "map points",
           lambda: pkg.search_projection_batch(prs), lambda: [O.search_projection(p) for p in prs], reps=10,
           algo_bytes=proj_bytes)
    # DBoW2 transform on a vocabulary of the size ORB-SLAM3 ships (k = 10, L = 6: 1 111 111 nodes, 35.5 MB), 1500 features,
    # levelsup = 4 (KeyFrame::ComputeBoW); byte model: L x k node descriptors per feature + the feature's own
    vocab = pkg.synth.make_vocabulary_full(2024, 10, 6)
    leaves = rng.integers(111111, 1111111, size=1500)
    bits = np.unpackbits(vocab["desc"][leaves], axis=1)
    bits ^= (rng.random(bits.shape) < 0.05).astype(np.uint8)
    dv = np.packbits(bits, axis=1)
    V = pkg.Vocabulary(vocab)
    report("DBoW2 transform, k=10 L=6 (1.1 M nodes), 1500 features, levelsup 4", 1500, "features", lambda: V.transform(dv, 4),
           lambda: O.vocab_transform(vocab, dv, 4), algo_bytes=1500 * (6 * 10 * 32 + 32 + 16))
    V.close()
    sizes = rng.integers(2, 25, size=4000)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    pool = rng.integers(0, 256, size=(int(offs[-1]), 32), dtype=np.uint8)
    report("ComputeDistinctiveDescriptors, 4000 map points", 4000, "map points",
           lambda: pkg.distinctive_descriptors(pool, offs), lambda: O.distinctive_descriptors(pool, offs))
    # Frame::ComputeStereoMatches on the pyramids the two extractors left on the device (EuRoC stereo, config 3)
    mbf, mb = 47.90639384423901, 47.90639384423901 / 435.2046959714599
    left, right = pkg.synth.make_stereo_pair(480, 752, 31, shift=24)
    exL, exR = pkg.ORBextractor(1200, 1.2, 8, 20, 7), pkg.ORBextractor(1200, 1.2, 8, 20, 7)
    (_, kL, dL), (_, kR, dR) = exL(left, (0, 0)), exR(right, (0, 0))
    oL, oR = O.Extractor(1200, 1.2, 8, 20, 7), O.Extractor(1200, 1.2, 8, 20, 7)
    (_, rkL, rdL), (_, rkR, rdR) = oL.extract(left, (0, 0)), oR.extract(right, (0, 0))
    g = timeit(lambda: pkg.compute_stereo_matches(exL, exR, kL, dL, kR, dR, mb, mbf), 20)
    c = timeit(lambda: O.compute_stereo_matches(oL, oR, rkL, rdL, rkR, rdR, mb, mbf), 4)
    print(json.dumps({"op": "ComputeStereoMatches 752x480 pair, %d x %d keypoints" % (len(kL), len(kR)), "units": len(kL),
                      "unit": "left keypoints", "call_ms": 1e3 * g, "cpu_oracle_ms_1core": 1e3 * c,
                      "call_units_per_s": len(kL) / g, "cpu_units_per_s": len(kL) / c}))
    gr = timeit(lambda: pkg.binding.compute_stereo_matches_resident(exL, exR, len(kL), mb, mbf), 50)
    print(json.dumps({"op": "ComputeStereoMatches, resident form (no keypoint / descriptor upload)", "units": len(kL),
                      "unit": "left keypoints", "call_ms": 1e3 * gr, "cpu_oracle_ms_1core": 1e3 * c}))
    # knn-2 on descriptors that are already on the device (the extractors' resident outputs)
    import torch
    _, pL, _, _, _ = exL.device_outputs()
    _, pR, _, _, _ = exR.device_outputs()
    d_idx = torch.zeros((len(kL), 2), dtype=torch.int32, device="cuda")
    d_dist = torch.zeros((len(kL), 2), dtype=torch.int32, device="cuda")

    def knn_dev():
        pkg.binding.bfknn2_device(pL, len(kL), pR, len(kR), d_idx.data_ptr(), d_dist.data_ptr())
        pkg.binding.matcher_sync()
    kd = timeit(knn_dev, 50)
    kh = timeit(lambda: pkg.bfknn2(dL, dR), 50)
    print(json.dumps({"op": "bfknn2 %d x %d: device-resident call + sync vs host-pointer call" % (len(kL), len(kR)),
                      "device_call_ms": 1e3 * kd, "host_call_ms": 1e3 * kh}))
    # cross-camera knn-2, 64 jobs of ~1000 x 1000 in one launch (K-KNN2F), device resident
    cap, frames = 1008, 64
    counts = rng.integers(900, cap, size=frames).astype(np.int32)
    desc = rng.integers(0, 256, size=(frames, cap, 32), dtype=np.uint8)
    dd, dc = torch.from_numpy(desc).cuda(), torch.from_numpy(counts).cuda()
    rec = np.zeros(frames, pkg.binding.KNN2_JOB_DTYPE)
    for k in range(frames):
        t = (k + 1) % frames
        rec[k] = (dd.data_ptr() + k * cap * 32, dc.data_ptr() + 4 * k, dd.data_ptr() + t * cap * 32, dc.data_ptr() + 4 * t)
    dj = torch.from_numpy(rec.view(np.uint8).copy()).cuda()
    di = torch.zeros((frames, cap, 2), dtype=torch.int32, device="cuda")
    ds = torch.zeros((frames, cap, 2), dtype=torch.int32, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = torch.cuda.Stream()  # (a NULL stream argument would mean "the calling thread's matcher stream")
    torch.cuda.synchronize()
    st = ts.cuda_stream
    pkg.binding.bfknn2_frames_device(dj.data_ptr(), frames, cap, di.data_ptr(), ds.data_ptr(), stream=st)
    e0.record(ts)
    for _ in range(20):
        pkg.binding.bfknn2_frames_device(dj.data_ptr(), frames, cap, di.data_ptr(), ds.data_ptr(), stream=st)
    e1.record(ts)
    torch.cuda.synchronize()
    kms = e0.elapsed_time(e1) / 20
    nd = float((counts.astype(np.float64) * counts[(np.arange(frames) + 1) % frames]).sum())
    kb = float(2 * counts.sum() * 32 + counts.sum() * 16)
    print(json.dumps({"op": "bfknn2_frames x64 (K-KNN2F, cross-camera ring)", "units": nd, "unit": "distances", "kernel_ms": kms,
                      "kernel_units_per_s": nd / (kms * 1e-3), "algorithmic_bytes": kb,
                      "kernel_GBps_algorithmic": kb / (kms * 1e-3) / 1e9, "frac_of_hbm_peak": kb / (kms * 1e-3) / 1e9 / 8000.0,
                      "note": "72 KB of operands per job: the bound of this kernel is the popcount rate, not HBM"}), flush=True)
    print(json.dumps({"op": "projection sweeps", "mode0": None, "last": pkg.search_projection_last_sweeps()}))


if __name__ == "__main__":
    main()
