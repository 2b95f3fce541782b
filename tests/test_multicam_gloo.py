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
