"""Extract -> match without leaving the device: the matcher reads the descriptors where the extractor (or the
all-gather) left them in HBM.  Every result is compared with the oracle run on the downloaded descriptors."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import orb_slam3_detailed_comments_kor_amd as p
    return p


def test_cross_camera_knn2_on_the_gathered_slab(pkg, oracle):
    """Device-resident batch -> DescriptorExchange slab -> (world 1: local copy stands in for the all-gather) ->
    CrossCameraMatcher: knn-2 of every frame against the next two cameras of the ring, one launch."""
    import torch
    from orb_slam3_detailed_comments_kor_amd.multicam import CrossCameraMatcher, PipelinedExchange, ring_pairs
    dev = torch.device("cuda", 0)
    B, H, W, nf = 5, 240, 376, 500
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ex.set_stream(stream.cuda_stream)
        cap = ex.max_keypoints(H, W)
        imgs = np.stack([pkg.synth.make_frame(H, W, 300 + i) if i != 3 else np.zeros((H, W), np.uint8) for i in range(B)])
        d_img = torch.from_numpy(imgs).pin_memory().to(dev)
        pipe = PipelinedExchange(B, cap, dev, 1, 0)
        d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
        d_mono = torch.zeros(B, dtype=torch.int32, device=dev)
        pairs = ring_pairs(1, B, 0, hops=(1, 2))
        cm = CrossCameraMatcher(pipe.x, pairs, dev)
        for rep in range(3):  # both slab pairs get used
            x = pipe.begin()
            ex.extract_batch_device(d_img.data_ptr(), B, H, W, W, H * W, (0, 0), d_kps.data_ptr(),
                                    x.desc_view().data_ptr(), cap, x.count_view().data_ptr(), d_mono.data_ptr())
            pipe.submit()
            pipe.drain()
            idx, dist = cm.match(x)
            torch.cuda.synchronize()
            n = x.count_view().cpu().numpy()
            desc = x.desc_view().cpu().numpy()
            assert n[3] == 0 and (n[[0, 1, 2, 4]] > 100).all()   # frame 3 is blank: empty query AND empty train frame
            idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
            for k, (q, g) in enumerate(pairs):
                ri, rd = oracle.bfknn2(desc[q, :n[q]], desc[g, :n[g]])
                assert np.array_equal(idx[k, :n[q]], ri) and np.array_equal(dist[k, :n[q]], rd), (rep, k)
    ex.close()


def test_matcher_reads_the_extractors_resident_output(pkg, oracle):
    """Host-pointer extraction, then DescriptorDistance / knn-2 / SearchByBoW on the descriptors the context still
    holds in HBM (orbfe_get_device_outputs): device forms and the device-pointer recognition of the host forms."""
    import torch
    left, right = pkg.synth.make_stereo_pair(240, 376, 77, shift=9)
    exL = pkg.ORBextractor(600, 1.2, 8, 20, 7)
    exR = pkg.ORBextractor(600, 1.2, 8, 20, 7)
    _, kL, dL = exL(left, (0, 0))
    _, kR, dR = exR(right, (0, 0))
    _, pL, _, _, _ = exL.device_outputs()
    _, pR, _, _, _ = exR.device_outputs()
    nL, nR = len(kL), len(kR)
    dev = torch.device("cuda", 0)
    d_idx = torch.zeros((nL, 2), dtype=torch.int32, device=dev)
    d_dist = torch.zeros((nL, 2), dtype=torch.int32, device=dev)
    d_D = torch.zeros((nL, nR), dtype=torch.int16, device=dev)
    torch.cuda.synchronize()
    pkg.binding.bfknn2_device(pL, nL, pR, nR, d_idx.data_ptr(), d_dist.data_ptr())
    pkg.binding.hamming_pairs_device(pL, nL, pR, nR, d_D.data_ptr())
    pkg.binding.matcher_sync()
    ri, rd = oracle.bfknn2(dL, dR)
    assert np.array_equal(d_idx.cpu().numpy(), ri) and np.array_equal(d_dist.cpu().numpy(), rd)
    assert np.array_equal(d_D.cpu().numpy().astype(np.uint16), oracle.hamming_matrix(dL, dR))
    # host entry points given device descriptor pointers
    idx2, dist2 = np.zeros((nL, 2), np.int32), np.zeros((nL, 2), np.int32)
    r = pkg.lib().orbfe_bfknn2(0, pL, nL, pR, nR, idx2.ctypes.data, dist2.ctypes.data)
    assert r == 0 and np.array_equal(idx2, ri) and np.array_equal(dist2, rd)
    fvL = pkg.synth.make_feature_vectors(dL, 7, 10, 2)
    fvR = pkg.synth.make_feature_vectors(dR, 7, 10, 2)
    mask = (np.arange(nL) % 5 != 0).astype(np.uint8)
    rn, rm = oracle.search_bow_kf_f(dL, mask, kL["angle"], fvL, dR, kR["angle"], fvR, -1, 0.7, True)
    for d1, d2 in (((pL, nL), (pR, nR)), ((pL, nL), dR), (dL, (pR, nR))):
        n, m = pkg.search_bow(d1, mask, kL["angle"], fvL, d2, None, kR["angle"], fvR, 0, 0.7, True)
        assert n == rn and np.array_equal(m, rm)
    out = pkg.search_bow_batch([dict(desc1=(pL, nL), mask1=mask, ang1=kL["angle"], fv1=fvL, desc2=(pR, nR), ang2=kR["angle"],
                                     fv2=fvR, variant=0, nnratio=0.7),
                                dict(desc1=dR, mask1=np.ones(nR, np.uint8), ang1=kR["angle"], fv1=fvR, desc2=(pL, nL),
                                     ang2=kL["angle"], fv2=fvL, variant=0, nnratio=0.8)])
    assert out[0][0] == rn and np.array_equal(out[0][1], rm)
    rn2, rm2 = oracle.search_bow_kf_f(dR, np.ones(nR, np.uint8), kR["angle"], fvR, dL, kL["angle"], fvL, -1, 0.8, True)
    assert out[1][0] == rn2 and np.array_equal(out[1][1], rm2)
    exL.close()
    exR.close()


def test_frames_knn2_ragged_counts_and_both_kernel_shapes(pkg, oracle):
    """Random descriptors, ragged counts incl. 0 / 1 / cap; few jobs (8 wavefronts per workgroup) and many (4)."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    for frames, cap, hops in ((3, 70, (1,)), (40, 130, (1, 2, 3, 5))):
        counts = rng.integers(0, cap + 1, size=frames).astype(np.int32)
        counts[0], counts[1], counts[2] = cap, 1, 0
        desc = rng.integers(0, 256, size=(frames, cap, 32), dtype=np.uint8)
        desc[1, 0] = desc[0, 5]  # exact duplicate -> distance 0, ties elsewhere from the small sets
        d_desc = torch.from_numpy(desc).pin_memory().to(dev)
        d_cnt = torch.from_numpy(counts).pin_memory().to(dev)
        pairs = [(i, (i + h) % frames) for h in hops for i in range(frames)]
        rec = np.zeros(len(pairs), pkg.binding.KNN2_JOB_DTYPE)
        for k, (q, t) in enumerate(pairs):
            rec[k] = (d_desc.data_ptr() + q * cap * 32, d_cnt.data_ptr() + 4 * q, d_desc.data_ptr() + t * cap * 32,
                      d_cnt.data_ptr() + 4 * t)
        d_jobs = torch.from_numpy(rec.view(np.uint8).copy()).pin_memory().to(dev)
        d_idx = torch.full((len(pairs), cap, 2), -7, dtype=torch.int32, device=dev)
        d_dist = torch.full((len(pairs), cap, 2), -7, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        pkg.binding.bfknn2_frames_device(d_jobs.data_ptr(), len(pairs), cap, d_idx.data_ptr(), d_dist.data_ptr())
        pkg.binding.matcher_sync()
        idx, dist = d_idx.cpu().numpy(), d_dist.cpu().numpy()
        for k, (q, t) in enumerate(pairs):
            ri, rd = oracle.bfknn2(desc[q, :counts[q]], desc[t, :counts[t]])
            assert np.array_equal(idx[k, :counts[q]], ri) and np.array_equal(dist[k, :counts[q]], rd), (frames, k)
            assert (idx[k, counts[q]:] == -7).all()  # rows beyond the query count are untouched


def _frames_knn2(pkg, torch, dev, desc, counts, pairs, cap):
    d_desc = torch.from_numpy(desc).pin_memory().to(dev)
    d_cnt = torch.from_numpy(counts).pin_memory().to(dev)
    rec = np.zeros(len(pairs), pkg.binding.KNN2_JOB_DTYPE)
    for k, (q, t) in enumerate(pairs):
        rec[k] = (d_desc.data_ptr() + q * cap * 32, d_cnt.data_ptr() + 4 * q, d_desc.data_ptr() + t * cap * 32, d_cnt.data_ptr() + 4 * t)
    d_jobs = torch.from_numpy(rec.view(np.uint8).copy()).pin_memory().to(dev)
    d_idx = torch.full((len(pairs), cap, 2), -7, dtype=torch.int32, device=dev)
    d_dist = torch.full((len(pairs), cap, 2), -7, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    pkg.binding.bfknn2_frames_device(d_jobs.data_ptr(), len(pairs), cap, d_idx.data_ptr(), d_dist.data_ptr())
    pkg.binding.matcher_sync()
    return d_idx.cpu().numpy(), d_dist.cpu().numpy()


def test_frames_knn2_on_the_matrix_pipe_is_exact(pkg, oracle):
    """k_bfknn2_frames_mfma (round 5): the knn-2 of many frame pairs as i8 MFMAs whose accumulator IS the scan's key.  Exact
    means exact: distances, indices and TIE ORDER (lower train index first, also for the second best) against the oracle's
    sequential scan, on (a) train counts around every tile boundary (0, 1, 2, 31, 32, 33, 63, 64, 65, cap), (b) descriptor sets
    drawn from a handful of prototypes so that most distances tie, (c) all-zero / all-one descriptors (distance 0 and 256, the
    ends of the key's range), (d) the largest frame the 11 index bits allow (2048), (e) query counts that leave the last
    workgroup / wavefront / tile partly empty, (f) job counts that are and are not multiples of 8 (the XCD-affine order)."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(2025)
    # (a) + (e) + (f): boundaries
    cap = 300
    tcounts = [0, 1, 2, 31, 32, 33, 63, 64, 65, 255, 256, 257, cap, 100, 7, 290]
    frames = len(tcounts)
    counts = np.array(tcounts, np.int32)
    desc = rng.integers(0, 256, size=(frames, cap, 32), dtype=np.uint8)
    pairs = [(q, t) for q in (12, 11, 9, 5, 1, 0) for t in range(frames)]  # 96 jobs: a multiple of 8
    idx, dist = _frames_knn2(pkg, torch, dev, desc, counts, pairs, cap)
    for k, (q, t) in enumerate(pairs):
        ri, rd = oracle.bfknn2(desc[q, :counts[q]], desc[t, :counts[t]])
        assert np.array_equal(idx[k, :counts[q]], ri) and np.array_equal(dist[k, :counts[q]], rd), ("boundaries", q, t)
        assert (idx[k, counts[q]:] == -7).all()
    pairs = pairs[:13]  # not a multiple of 8
    idx, dist = _frames_knn2(pkg, torch, dev, desc, counts, pairs, cap)
    for k, (q, t) in enumerate(pairs):
        ri, rd = oracle.bfknn2(desc[q, :counts[q]], desc[t, :counts[t]])
        assert np.array_equal(idx[k, :counts[q]], ri) and np.array_equal(dist[k, :counts[q]], rd), ("13 jobs", q, t)
    # (b) + (c): ties everywhere
    cap = 520
    proto = rng.integers(0, 256, size=(6, 32), dtype=np.uint8)
    proto[0] = 0
    proto[1] = 255
    frames = 8
    desc = proto[rng.integers(0, 6, size=(frames, cap))]
    flip = rng.random((frames, cap)) < 0.3  # a third of the rows one bit away from their prototype
    desc[flip, 7] ^= 0x10
    counts = rng.integers(400, cap + 1, size=frames).astype(np.int32)
    pairs = [(i, (i + 3) % frames) for i in range(frames)]
    idx, dist = _frames_knn2(pkg, torch, dev, desc, counts, pairs, cap)
    for k, (q, t) in enumerate(pairs):
        ri, rd = oracle.bfknn2(desc[q, :counts[q]], desc[t, :counts[t]])
        assert np.array_equal(idx[k, :counts[q]], ri) and np.array_equal(dist[k, :counts[q]], rd), ("ties", q, t)
        assert (rd[:, 0] == rd[:, 1]).mean() > 0.5  # (the case really is about ties)
    # (d): 2048 rows -- the index field full -- and just above it (the vector-pipe kernel takes over: keys of 20 index bits)
    for cap in (2048, 2100):
        frames = 3
        counts = np.array([cap, cap - 37, 1999], np.int32)
        desc = rng.integers(0, 256, size=(frames, cap, 32), dtype=np.uint8)
        desc[1, cap - 40] = desc[0, 2047]  # a best match at the very last index of the field
        pairs = [(0, 1), (1, 0), (2, 0), (1, 2)]
        idx, dist = _frames_knn2(pkg, torch, dev, desc, counts, pairs, cap)
        for k, (q, t) in enumerate(pairs):
            ri, rd = oracle.bfknn2(desc[q, :counts[q]], desc[t, :counts[t]])
            assert np.array_equal(idx[k, :counts[q]], ri) and np.array_equal(dist[k, :counts[q]], rd), (cap, q, t)


def test_matcher_orders_itself_after_an_asynchronous_extraction(pkg, oracle):
    """orbfe_extract_batch_device returns at once; orbfe_get_device_outputs marks the context's stream and a matcher call
    that is handed the resident descriptors waits for that mark on its own stream -- no orbfe_sync in between (the calls
    used to race).  A large batch in front makes the window wide."""
    import torch
    dev = torch.device("cuda", 0)
    B, H, W, nf = 24, 480, 752, 1000
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    cap = ex.max_keypoints(H, W)
    imgs = np.stack([pkg.synth.make_frame(H, W, 40 + i) for i in range(B)])
    d_img = torch.from_numpy(imgs).pin_memory().to(dev)
    d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    d_mono = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    refs = [oracle.Extractor(nf, 1.2, 8, 20, 7).extract(imgs[i], (0, 0)) for i in (0, B - 1)]
    n0, n1 = len(refs[0][1]), len(refs[1][1])
    ri, rd = oracle.bfknn2(refs[0][2], refs[1][2])
    for rep in range(3):
        d_desc.zero_()
        torch.cuda.synchronize()
        ex.extract_batch_device(d_img.data_ptr(), B, H, W, W, H * W, (0, 0), d_kps.data_ptr(), d_desc.data_ptr(), cap,
                                d_n.data_ptr(), d_mono.data_ptr())
        _, p_desc, _, pcap, pn = ex.device_outputs()   # no sync: the batch is still running
        assert (pcap, pn) == (cap, B) and p_desc == d_desc.data_ptr()
        idx, dist = np.zeros((n0, 2), np.int32), np.zeros((n0, 2), np.int32)
        r = pkg.lib().orbfe_bfknn2(0, p_desc, n0, p_desc + (B - 1) * cap * 32, n1, idx.ctypes.data, dist.ctypes.data)
        assert r == 0 and np.array_equal(idx, ri) and np.array_equal(dist, rd), rep
    ex.close()
