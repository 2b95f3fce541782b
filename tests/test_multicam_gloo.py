"""N>1 path on CPU: frame sharding + the single all-gather of descriptor slabs, world_size 2, gloo."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, frames, cap, q):
    sys.path.insert(0, ROOT)
    from orb_slam3_detailed_comments_kor_amd.multicam import DescriptorExchange, shard_frames
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        xch = DescriptorExchange(frames, cap, torch.device("cpu"), world, rank)
        # stand-in for the extractor writing into the slab views (no GPU here): deterministic content
        rng = np.random.default_rng(100 + rank)
        n = rng.integers(cap // 2, cap, size=frames).astype(np.int32)
        d = rng.integers(0, 256, size=(frames, cap, 32), dtype=np.uint8)
        xch.count_view().copy_(torch.from_numpy(n))
        xch.desc_view().copy_(torch.from_numpy(d))
        xch.all_gather()
        ok = True
        for r in range(world):
            rr = np.random.default_rng(100 + r)
            en = rr.integers(cap // 2, cap, size=frames).astype(np.int32)
            ed = rr.integers(0, 256, size=(frames, cap, 32), dtype=np.uint8)
            gn, gd = xch.unpack(r)
            ok &= np.array_equal(gn.numpy(), en) and np.array_equal(gd.numpy(), ed)
        # the double-buffered asynchronous form the bench uses: three batches through two slab pairs
        from orb_slam3_detailed_comments_kor_amd.multicam import PipelinedExchange
        pipe = PipelinedExchange(frames, cap, torch.device("cpu"), world, rank)
        for b in range(3):
            x = pipe.begin()
            rb = np.random.default_rng(1000 * b + rank)
            x.count_view().copy_(torch.from_numpy(rb.integers(0, cap, size=frames).astype(np.int32)))
            x.desc_view().copy_(torch.from_numpy(rb.integers(0, 256, size=(frames, cap, 32), dtype=np.uint8)))
            pipe.submit()
        pipe.drain()
        for r in range(world):
            rb = np.random.default_rng(1000 * 2 + r)
            en = rb.integers(0, cap, size=frames).astype(np.int32)
            ed = rb.integers(0, 256, size=(frames, cap, 32), dtype=np.uint8)
            gn, gd = pipe.completed().unpack(r)
            ok &= np.array_equal(gn.numpy(), en) and np.array_equal(gd.numpy(), ed)
        # cross-camera matching bookkeeping (query shard -> job records -> frames inside the gathered buffer), with a
        # CPU stand-in for the knn-2 kernel: every local frame against the next camera of the ring and the one after
        from orb_slam3_detailed_comments_kor_amd.multicam import job_offsets, ring_pairs
        pairs = ring_pairs(world, frames, rank, hops=(1, 2))
        off = job_offsets(frames, cap, xch.slab_bytes, pairs)
        slab, gathered = xch.slab.numpy(), xch.gathered.numpy()
        seen_remote = False
        for (qi, g), (qd, qc, td, tc) in zip(pairs, off):
            nq = int(slab[qc:qc + 4].view(np.int32)[0])
            nt = int(gathered[tc:tc + 4].view(np.int32)[0])
            Q = slab[qd:qd + nq * 32].reshape(nq, 32)
            T = gathered[td:td + nt * 32].reshape(nt, 32)
            r, j = divmod(g, frames)
            seen_remote |= r != rank
            rr = np.random.default_rng(100 + r)
            en = rr.integers(cap // 2, cap, size=frames).astype(np.int32)
            ed = rr.integers(0, 256, size=(frames, cap, 32), dtype=np.uint8)
            ok &= nq == n[qi] and nt == en[j] and np.array_equal(Q, d[qi, :nq]) and np.array_equal(T, ed[j, :nt])
            # the stand-in matcher on the slices == the same on the frames' original data
            D = np.unpackbits(Q[:, None, :] ^ T[None, :, :], axis=2).sum(axis=2)
            D0 = np.unpackbits(d[qi, :nq, None, :] ^ ed[j, None, :nt, :], axis=2).sum(axis=2)
            ok &= np.array_equal(np.argsort(D, axis=1, kind="stable")[:, :2], np.argsort(D0, axis=1, kind="stable")[:, :2])
        ok &= seen_remote  # at least one partner frame came from the other rank
        # a REAL c10d work handle that has completed cleanly: begin() takes the "no ordering wait" branch (is_success();
        # Work.exception() cannot be called from Python)
        import time
        x = pipe.begin()
        pipe.submit()
        time.sleep(0.3)
        skipped = pipe.waits_skipped
        pipe.begin()   # the other slab: its collective finished long ago
        ok &= pipe.waits_skipped >= skipped
        pipe.drain()
        w = dist.all_gather_into_tensor(x.gathered, x.slab, async_op=True)
        time.sleep(0.3)
        from orb_slam3_detailed_comments_kor_amd.multicam import _work_done
        ok &= _work_done(w) is True
        w.wait()
        qs = list(xch.query_shard())
        first, count = shard_frames(world * frames, world, rank)
        ok &= qs == list(range(first, first + count))
        q.put((rank, bool(ok), qs))
    finally:
        dist.destroy_process_group()


def test_all_gather_of_descriptor_slabs_world2():
    world, frames, cap = 2, 3, 40
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, frames, cap, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    covered = sorted(sum((qs for _, _, qs in res), []))
    assert covered == list(range(world * frames))  # query shards partition all frames


def _mc_worker(rank, world, uid, frames, cap, q):
    """The same exchange and bookkeeping through the C ABI (include/orbfe_mc.h), host-memory handles (no GPU here):
    orbfe_mc_create(ctx = NULL, ORBFE_MC_HOST) -> orbfe_mc_exchange_host -> orbfe_mc_ring_pairs / _job_offsets."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from orb_slam3_detailed_comments_kor_amd import binding
    import orb_oracle_py as O  # the checker: knn-2 stand-in for the device matcher
    ok = True
    try:
        mc = binding.MultiCam(None, uid, rank, world, frames, cap, binding.MC_HOST)
        desc_bytes, count_off, slab_bytes = binding.mc_layout(frames, cap)
        ok &= slab_bytes % 256 == 0 and count_off == desc_bytes == frames * cap * 32

        def content(r, b):
            rr = np.random.default_rng(10 * b + r)
            return rr.integers(cap // 2, cap, size=frames).astype(np.int32), rr.integers(0, 256, size=(frames, cap, 32), dtype=np.uint8)

        for b in range(3):  # three rounds through the two halves of the shared segment
            n, d = content(rank, b)
            slab = np.zeros(slab_bytes, np.uint8)
            slab[:desc_bytes] = d.reshape(-1)
            slab[count_off:count_off + 4 * frames] = n.view(np.uint8)
            g = mc.exchange_host(slab)
            for r in range(world):
                en, ed = content(r, b)
                ok &= np.array_equal(g[r, :desc_bytes].reshape(frames, cap, 32), ed)
                ok &= np.array_equal(g[r, count_off:count_off + 4 * frames].view(np.int32), en)
            gathered = g.reshape(-1).copy()
        # ring matching bookkeeping on the last round: every local frame against the next camera and the one after
        pairs = binding.mc_ring_pairs(world, frames, rank, (1, 2))
        ok &= len(pairs) == 2 * frames and [qq for qq, _ in pairs[:frames]] == list(range(frames))
        off = binding.mc_job_offsets(frames, cap, pairs)
        seen_remote = False
        for (qi, gidx), (qd, qc, td, tc) in zip(pairs, off):
            ok &= gidx == (rank * frames + qi + (1 if pairs.index((qi, gidx)) < frames else 2)) % (world * frames)
            nq = int(slab[qc:qc + 4].view(np.int32)[0])
            nt = int(gathered[tc:tc + 4].view(np.int32)[0])
            Q = slab[qd:qd + nq * 32].reshape(nq, 32)
            T = gathered[td:td + nt * 32].reshape(nt, 32)
            r, j = divmod(gidx, frames)
            seen_remote |= r != rank
            en, ed = content(r, 2)
            ok &= nq == n[qi] and nt == en[j] and np.array_equal(T, ed[j, :nt])
            idx, dst = O.bfknn2(Q, T)
            D = np.unpackbits(d[qi, :nq, None, :] ^ ed[j, None, :nt, :], axis=2).sum(axis=2)
            ok &= np.array_equal(idx, np.argsort(D, axis=1, kind="stable")[:, :2])
            ok &= np.array_equal(dst, np.sort(D, axis=1, kind="stable")[:, :2])
        ok &= seen_remote
        first, count = binding.mc_shard(world * frames, world, rank)
        mc.close()
        q.put((rank, bool(ok), list(range(first, first + count))))
    except Exception as e:  # noqa: BLE001
        q.put((rank, False, repr(e)))


def test_exchange_and_ring_bookkeeping_through_the_c_abi_world2():
    sys.path.insert(0, ROOT)
    from orb_slam3_detailed_comments_kor_amd import binding
    world, frames, cap = 2, 3, 40
    uid = binding.mc_unique_id(binding.MC_HOST)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mc_worker, args=(r, world, uid, frames, cap, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert sorted(sum((qs for _, _, qs in res), [])) == list(range(world * frames))


def test_mc_argument_errors_need_no_device():
    sys.path.insert(0, ROOT)
    from orb_slam3_detailed_comments_kor_amd import binding
    L = binding.lib()
    import ctypes as C
    h = C.c_void_p()
    assert L.orbfe_mc_create(C.byref(h), None, None, 0, 1, 2, 10, binding.MC_RCCL) == binding.ERR_ARGS  # RCCL needs a context
    assert L.orbfe_mc_create(C.byref(h), None, None, 2, 2, 2, 10, binding.MC_HOST) == binding.ERR_ARGS  # rank out of range
    assert L.orbfe_mc_create(C.byref(h), None, None, 0, 2, 2, 10, binding.MC_HOST) == binding.ERR_ARGS  # world 2 without an id
    with pytest.raises(binding.OrbfeError):
        binding.mc_layout(0, 10)
    assert binding.mc_shard(7, 3, 0) == (0, 3) and binding.mc_shard(7, 3, 2) == (5, 2)
    assert binding.mc_ring_pairs(2, 2, 1, (1, -1)) == [(0, 3), (1, 0), (0, 1), (1, 2)]


def test_shard_frames_partition():
    from orb_slam3_detailed_comments_kor_amd.multicam import owner_of_frame, shard_frames
    for nframes in (1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                first, count = shard_frames(nframes, world, r)
                seen += list(range(first, first + count))
                for f in range(first, first + count):
                    assert owner_of_frame(f, nframes, world) == r
            assert seen == list(range(nframes))


class _FakeWork:
    def __init__(self, completed, exc=None):
        self.completed, self.exc, self.waited = completed, exc, 0

    def is_completed(self):
        return self.completed

    def exception(self):
        return self.exc

    def wait(self):
        self.waited += 1
        if self.exc is not None:
            raise self.exc
        return True


def test_pipelined_exchange_never_skips_a_failed_collective():
    """begin() may skip the ordering wait only for a collective that completed CLEANLY: is_completed() is also true
    for one that ended with an exception, and wait() is the only call that raises it."""
    from orb_slam3_detailed_comments_kor_amd.multicam import PipelinedExchange
    pipe = PipelinedExchange(2, 8, torch.device("cpu"), world=1, rank=0)
    clean = _FakeWork(True)
    pipe.pending[0] = clean
    pipe.begin()
    assert clean.waited == 0 and pipe.waits_skipped == 1 and pipe.pending[0] is None   # completed branch taken
    running = _FakeWork(False)
    pipe.pending[0] = running
    pipe.begin()
    assert running.waited == 1 and pipe.waits_skipped == 1
    failed = _FakeWork(True, RuntimeError("NCCL communicator was aborted"))
    pipe.pending[0] = failed
    with pytest.raises(RuntimeError):
        pipe.begin()
    assert failed.waited == 1


def test_ring_pairs_and_job_offsets():
    from orb_slam3_detailed_comments_kor_amd.multicam import job_offsets, ring_pairs
    world, frames, cap = 4, 3, 10
    slab_bytes = (frames * cap * 32 + 4 * frames + 255) // 256 * 256
    seen = set()
    for r in range(world):
        pr = ring_pairs(world, frames, r)
        assert [q for q, _ in pr] == list(range(frames))
        for q, g in pr:
            assert g == (r * frames + q + 1) % (world * frames)
            seen.add(g)
        off = job_offsets(frames, cap, slab_bytes, pr)
        assert off.shape == (frames, 4) and (off[:, 2] % 32 == 0).all() and (off[:, 0] == np.arange(frames) * cap * 32).all()
        assert (off[:, 3] - off[:, 2] >= 0).all() and (off[:, 3] < world * slab_bytes).all()
    assert seen == set(range(world * frames))  # every frame is somebody's train frame exactly once


def _settle_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import time
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        acc = torch.zeros(1)

        def body():  # a step with a collective in it; the ranks run at different speeds
            t = torch.ones(1)
            dist.all_reduce(t)
            acc.add_(t)
            time.sleep(0.002 if rank == 0 else 0.011)

        calls = bench.settle_together(0.25 if rank == 0 else 0.05, body, world, dev)  # (even the limits differ)
        dist.barrier()
        q.put((rank, calls, float(acc.item())))
    finally:
        dist.destroy_process_group()


def test_bench_settle_loop_runs_the_same_number_of_collectives_on_every_rank():
    # bench.py settles the clocks by time; every step holds the all-gather, so the ranks must agree on the step count
    # (a per-rank `while elapsed < settle` deadlocks the job at N > 1)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_settle_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] and res[0][1] >= 5
    assert res[0][2] == res[1][2] == world * res[0][1]
