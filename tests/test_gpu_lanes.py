"""Lanes (orbfe_set_lanes / orbfe_set_lane_mode, include/orbfe.h): 2..4 device-pointer batches in flight on streams the context
owns -- whole batches round-robin.
Outputs must be bit-identical to the one-lane results (and to the oracle's), and every documented join point must really order
the lanes: orbfe_sync, orbfe_lanes_join + work on the context's stream, orbfe_get_device_outputs + a matcher call,
orbfe_get_level, a host-pointer call, a batch of another size.  Batch lanes in addition: a ring of output sets with DIFFERENT
inputs per call and no join between the calls, the SAME image buffer refilled on the context's stream between calls (the
input guard: ADVICE r04), shapes that change while lanes are busy, and all of it under PCIe load from other host threads."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("x", "y", "size", "angle", "response", "octave", "class_id")


@pytest.fixture(scope="module")
def pkg():
    import orb_slam3_detailed_comments_kor_amd as p
    return p


def _bufs(torch, B, cap, dev):
    return (torch.zeros((B, cap, 7), dtype=torch.float32, device=dev), torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev),
            torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev))


def _teq(torch, a, b):
    # (bitwise: a keypoint's class_id = -1 reads as NaN in the float view of the records)
    if a.dtype == torch.float32:
        return torch.equal(a.view(torch.int32), b.view(torch.int32))
    return torch.equal(a, b)


def _run(ex, d_img, B, H, W, lap, out, cap, zero_on=None):
    if zero_on is not None:  # rows past n[i] are unspecified: clear the arrays ON the context's stream first (which also checks
        import torch         # that the second lane does not start before this point of the stream)
        with torch.cuda.stream(zero_on):
            for t in out:
                t.zero_()
    ex.extract_batch_device(d_img.data_ptr(), B, H, W, W, H * W, lap, out[0].data_ptr(), out[1].data_ptr(), cap,
                            out[2].data_ptr(), out[3].data_ptr())


def _check_against_oracle(pkg, oracle, imgs, lap, out, nf, idxs):
    n = out[2].cpu().numpy()
    mono = out[3].cpu().numpy()
    kps = out[0].cpu().numpy()
    desc = out[1].cpu().numpy()
    ref = oracle.Extractor(nf, 1.2, 8, 20, 7)
    for i in idxs:
        rmono, rkps, rdesc = ref.extract(imgs[i], lap)
        assert n[i] == len(rkps) and mono[i] == rmono, i
        k = kps[i, : n[i]].copy().view(pkg.KP_DTYPE).reshape(-1)
        for f in FIELDS:
            assert np.array_equal(k[f], rkps[f]), (i, f)
        assert np.array_equal(desc[i, : n[i]], rdesc), i


MODES = [(2, 0), (3, 0), (4, 0)]  # (lanes, mode): ORBFE_LANES_BATCH is the one mode (round 4's half-batches left in round 6)


@pytest.mark.parametrize("lanes,mode", MODES)
@pytest.mark.parametrize("B,hw", [(64, (300, 500)), (24, (480, 752)), (17, (240, 376))])
def test_lanes_equal_one_lane_and_the_oracle(pkg, oracle, B, hw, lanes, mode):
    import torch
    H, W = hw
    nf = 800
    dev = torch.device("cuda:0")
    kinds = list(pkg.synth.FRAME_KINDS)
    imgs = np.stack([pkg.synth.make_frame_kind(H, W, 500 + i, kinds[i % len(kinds)]) for i in range(B)])
    d_img = torch.from_numpy(imgs).pin_memory().to(dev)
    ex1 = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    ex2 = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    ex2.set_lanes(lanes, mode)
    cap = ex1.max_keypoints(H, W)
    o1, o2 = _bufs(torch, B, cap, dev), _bufs(torch, B, cap, dev)
    torch.cuda.synchronize()
    lap = (100, 400)
    _run(ex1, d_img, B, H, W, lap, o1, cap)
    ex1.sync()
    for _ in range(5):  # consecutive calls: both lanes free-running
        _run(ex2, d_img, B, H, W, lap, o2, cap)
    ex2.sync()
    for a, b in zip(o1, o2):
        assert _teq(torch, a, b)
    _check_against_oracle(pkg, oracle, imgs, lap, o2, nf, sorted({0, B // 2 - 1, B // 2, B - 1, 8, 9}))
    ex1.close()
    ex2.close()


@pytest.mark.parametrize("lanes,mode", MODES)
def test_join_points_order_the_lanes(pkg, oracle, lanes, mode):
    import torch
    B, H, W, nf = 32, 300, 500, 800
    dev = torch.device("cuda:0")
    imgs = np.stack([pkg.synth.make_frame(H, W, 800 + i) for i in range(B)])
    imgs2 = np.stack([pkg.synth.make_frame(H, W, 900 + i) for i in range(B)])
    d_a, d_b = torch.from_numpy(imgs).pin_memory().to(dev), torch.from_numpy(imgs2).pin_memory().to(dev)
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    ex.set_lanes(lanes, mode)
    stream = torch.cuda.Stream(device=dev)
    ex.set_stream(stream.cuda_stream)
    cap = ex.max_keypoints(H, W)
    out = _bufs(torch, B, cap, dev)
    ref1 = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    want_a, want_b = _bufs(torch, B, cap, dev), _bufs(torch, B, cap, dev)
    _run(ref1, d_a, B, H, W, (0, 0), want_a, cap)
    _run(ref1, d_b, B, H, W, (0, 0), want_b, cap)
    ref1.sync()
    torch.cuda.synchronize()
    # (1) orbfe_lanes_join, then the caller's own work on the context's stream: a copy of the outputs taken ON that stream
    for rep in range(6):
        src, want = (d_a, want_a) if rep % 2 == 0 else (d_b, want_b)
        _run(ex, src, B, H, W, (0, 0), out, cap, zero_on=stream)
        ex.lanes_join()
        with torch.cuda.stream(stream):
            snap = [t.clone() for t in out]
        stream.synchronize()
        for a, b in zip(snap, want):
            assert _teq(torch, a, b), rep
    # (2) orbfe_get_device_outputs + a matcher call on the resident descriptors of the LAST image (second lane)
    _run(ex, d_a, B, H, W, (0, 0), out, cap)
    d_kps, d_desc, d_n, cap2, nimg = ex.device_outputs()
    assert nimg == B and cap2 == cap
    na = int(want_a[2][B - 1].item())
    d_idx = torch.full((na, 2), -1, dtype=torch.int32, device=dev)
    d_dist = torch.full((na, 2), -1, dtype=torch.int32, device=dev)
    last = d_desc + (B - 1) * cap * 32  # the last image's rows: written by the second lane
    pkg.binding.bfknn2_device(last, na, last, na, d_idx.data_ptr(), d_dist.data_ptr())
    pkg.binding.matcher_sync()
    hd = want_a[1][B - 1, :na].cpu().numpy()
    ridx, rdist = oracle.bfknn2(hd, hd)
    assert np.array_equal(d_idx.cpu().numpy(), ridx) and np.array_equal(d_dist.cpu().numpy(), rdist)
    # (3) orbfe_get_level of an image of the second half, right behind a call
    _run(ex, d_b, B, H, W, (0, 0), out, cap)
    lvl = ex.image_pyramid_level(2, img_index=B - 1)
    r = oracle.Extractor(nf, 1.2, 8, 20, 7)
    r.extract(imgs2[B - 1], (0, 0))
    assert np.array_equal(lvl, r.level(2))
    # (4) a host-pointer call and a batch of another size right behind a two-lane call
    _run(ex, d_a, B, H, W, (0, 0), out, cap)
    mono, kps, desc = ex(imgs2[3], (0, 0))
    rmono, rkps, rdesc = r.extract(imgs2[3], (0, 0))
    assert mono == rmono and np.array_equal(desc, rdesc)
    _run(ex, d_a, B, H, W, (0, 0), out, cap)
    out20 = _bufs(torch, 20, cap, dev)
    torch.cuda.synchronize()  # (out20 was zeroed on torch's current stream)
    _run(ex, d_b, 20, H, W, (0, 0), out20, cap)
    ex.sync()
    for a, b in zip(out20, want_b):
        assert _teq(torch, a, b[:20])
    # and back to one lane
    ex.set_lanes(1)
    _run(ex, d_a, B, H, W, (0, 0), out, cap, zero_on=stream)
    ex.sync()
    for a, b in zip(out, want_a):
        assert _teq(torch, a, b)
    ex.close()
    ref1.close()


def _oracle_outputs(oracle, imgs, lap, nf):
    ref = oracle.Extractor(nf, 1.2, 8, 20, 7)
    return [ref.extract(im, lap) for im in imgs]


def _check_set(pkg, out, want, tag):
    n, mono, kps, desc = out[2].cpu().numpy(), out[3].cpu().numpy(), out[0].cpu().numpy(), out[1].cpu().numpy()
    for i, (rmono, rkps, rdesc) in enumerate(want):
        assert n[i] == len(rkps) and mono[i] == rmono, (tag, i)
        k = kps[i, : n[i]].copy().view(pkg.KP_DTYPE).reshape(-1)
        for f in FIELDS:
            assert np.array_equal(k[f], rkps[f]), (tag, i, f)
        assert np.array_equal(desc[i, : n[i]], rdesc), (tag, i)


@pytest.mark.parametrize("lanes", [2, 3, 4])
def test_batch_lanes_ring_of_outputs_distinct_inputs_no_join(pkg, oracle, lanes):
    # n lanes, a ring of n output sets, twelve calls on twelve DIFFERENT small batches (the < 16-frame regime: 8 frames) queued
    # back to back without any join: call k must land in set k mod n with batch k's results; the last n sets are compared
    # after ONE orbfe_sync, the earlier ones are snapshotted behind orbfe_lanes_record-style joins every n calls
    import torch
    B, H, W, nf = 8, 300, 500, 700
    dev = torch.device("cuda:0")
    ncall = 12
    imgs = [np.stack([pkg.synth.make_frame_kind(H, W, 3000 + 16 * k + i, pkg.synth.FRAME_KINDS[(k + i) % 7]) for i in range(B)])
            for k in range(ncall)]
    d_imgs = [torch.from_numpy(a).pin_memory().to(dev) for a in imgs]
    lap = (120, 380)
    want = [_oracle_outputs(oracle, a, lap, nf) for a in imgs]
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    ex.set_lanes(lanes)
    cap = ex.max_keypoints(H, W)
    ring = [_bufs(torch, B, cap, dev) for _ in range(lanes)]
    torch.cuda.synchronize()
    for k in range(ncall):
        if k >= lanes and k % lanes == 0:  # the ring is about to be overwritten: check what it holds (calls k-lanes .. k-1)
            ex.sync()
            for j in range(lanes):
                _check_set(pkg, ring[j], want[k - lanes + j], "call %d" % (k - lanes + j))
        _run(ex, d_imgs[k], B, H, W, lap, ring[k % lanes], cap)
    ex.sync()
    for j in range(ncall - lanes, ncall):
        _check_set(pkg, ring[j % lanes], want[j], "call %d" % j)
    # "the last call" of the stage getters is call ncall-1, whatever lane it ran on
    r = oracle.Extractor(nf, 1.2, 8, 20, 7)
    r.extract(imgs[ncall - 1][B - 1], lap)
    assert np.array_equal(ex.image_pyramid_level(1, img_index=B - 1), r.level(1))
    ex.close()


def _load_threads(torch, dev, stop):
    """Two host threads saturating the link in both directions on streams of their own (tests/test_gpu_hostpath.py does the same
    to the latency paths)."""
    def h2d():
        s = torch.cuda.Stream(device=dev)
        src = torch.empty(8 << 20, dtype=torch.uint8).pin_memory()
        dst = torch.empty(8 << 20, dtype=torch.uint8, device=dev)
        with torch.cuda.stream(s):
            while not stop.is_set():
                dst.copy_(src, non_blocking=True)
                s.synchronize()

    def d2h():
        s = torch.cuda.Stream(device=dev)
        dst = torch.empty(8 << 20, dtype=torch.uint8).pin_memory()
        src = torch.empty(8 << 20, dtype=torch.uint8, device=dev)
        with torch.cuda.stream(s):
            while not stop.is_set():
                dst.copy_(src, non_blocking=True)
                s.synchronize()
    ts = [threading.Thread(target=h2d), threading.Thread(target=d2h)]
    for t in ts:
        t.start()
    return ts


@pytest.mark.parametrize("lanes", [2, 3, 4])
@pytest.mark.parametrize("guard_by", ["upload", "device_copy"])
def test_batch_lanes_same_image_buffer_refilled_between_calls(pkg, oracle, lanes, guard_by):
    # ONE image buffer, refilled ON THE CONTEXT'S STREAM right after every call returns -- by an upload from
    # pinned memory or by a device copy --, never a join, under PCIe load from two other host threads.  Safe by stream order
    # with one lane; with lanes the context's stream waits for the lane's pyramid kernel before anything queued behind the
    # call.  Every call's outputs must be its OWN batch's (a lane that read the buffer late would produce the NEXT batch's).
    import torch
    B, H, W, nf = 8, 240, 376, 500
    dev = torch.device("cuda:0")
    nb = 6
    imgs = [np.stack([pkg.synth.make_frame(H, W, 7000 + 8 * k + i) for i in range(B)]) for k in range(nb)]
    lap = (0, 1000)
    want = [_oracle_outputs(oracle, a, lap, nf) for a in imgs]
    pinned = [torch.from_numpy(a).pin_memory() for a in imgs]
    resident = [torch.from_numpy(a).pin_memory().to(dev) for a in imgs]
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    ex.set_lanes(lanes)
    stream = torch.cuda.Stream(device=dev)
    ex.set_stream(stream.cuda_stream)
    cap = ex.max_keypoints(H, W)
    ring = [_bufs(torch, B, cap, dev) for _ in range(lanes)]
    d_img = torch.zeros((B, H, W), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    stop = threading.Event()
    ts = _load_threads(torch, dev, stop)
    try:
        ncall = 60
        for k in range(ncall):
            with torch.cuda.stream(stream):  # the refill: ordered on the context's stream, nothing else
                d_img.copy_(pinned[k % nb] if guard_by == "upload" else resident[k % nb], non_blocking=True)
            if k >= lanes and k % lanes == 0:
                ex.sync()
                for j in range(lanes):
                    _check_set(pkg, ring[j], want[(k - lanes + j) % nb], "call %d" % (k - lanes + j))
            _run(ex, d_img, B, H, W, lap, ring[k % lanes], cap)
        ex.sync()
        for j in range(ncall - lanes, ncall):
            _check_set(pkg, ring[j % lanes], want[j % nb], "call %d" % j)
    finally:
        stop.set()
        for t in ts:
            t.join()
    ex.close()


def test_batch_lanes_shapes_change_while_lanes_are_busy(pkg, oracle):
    # calls of different image sizes and batch sizes follow each other with lanes busy: the size-dependent tables are swapped
    # under quiesced lanes, every lane re-checks its buffers; nF 1000 on the photographs' size joins in
    import torch
    import natural
    dev = torch.device("cuda:0")
    nf = 600
    shapes = [(8, 240, 376), (3, 300, 500), (8, 240, 376), (1, 427, 640), (12, 300, 500), (8, 240, 376), (2, 427, 640)]
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7)
    ex.set_lanes(3)
    outs, wants = [], []
    for k, (B, H, W) in enumerate(shapes):
        if (H, W) == (427, 640):
            imgs = np.stack([natural.random_frame(H, W, 40 + k + i) for i in range(B)])
        else:
            imgs = np.stack([pkg.synth.make_frame(H, W, 8100 + 16 * k + i) for i in range(B)])
        cap = ex.max_keypoints(H, W)
        out = _bufs(torch, B, cap, dev)
        d = torch.from_numpy(imgs).pin_memory().to(dev)
        torch.cuda.synchronize()
        _run(ex, d, B, H, W, (0, 0), out, cap)
        outs.append((out, d))
        wants.append(_oracle_outputs(oracle, imgs, (0, 0), nf))
    ex.sync()
    for k in range(len(shapes)):
        _check_set(pkg, outs[k][0], wants[k], "shape %d" % k)
    ex.close()
