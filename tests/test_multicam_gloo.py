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
