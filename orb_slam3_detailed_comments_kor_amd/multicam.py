"""Multi-camera / multi-frame sharding across the GPUs of one node.

Independent image pyramids shard naturally: frame f of a batch goes to one rank, each rank runs
the single-GPU batched extractor on its shard, and ONE all-gather of fixed-size descriptor slabs
(RCCL over xGMI when the backend is "nccl") gives every rank all descriptors for cross-camera
matching.  There is no reduction anywhere on this path.

Slab layout per rank (one contiguous uint8 tensor, so the exchange is a single collective):
    [ frames_per_rank * cap * 32 bytes of descriptors | frames_per_rank * int32 keypoint counts | pad to 256 B ]
The extractor writes straight into views of the slab, so no packing copy is needed.

After the gather every rank holds every camera's descriptors, and the cross-camera matching is sharded by QUERY
frame: `CrossCameraMatcher` runs, for each of this rank's frames, the knn-2 against its partner frames wherever in
the gathered buffer they live -- one launch (orbfe_bfknn2_frames_device) on job records that name both frames by
device address, so nothing is repacked after the collective.

Round 3: the path lives behind the C ABI (include/orbfe_mc.h, csrc/orbfe_multicam.hip: slab layout, sharding, ring
pairs, job records, the double-buffered extraction + ncclAllGather on a side stream, the ring matching) so that a C++
host reaches it; `binding.MultiCam` is its ctypes handle and what bench.py runs at N > 1.  The functions below are thin
calls into that ABI; the torch.distributed classes further down are the same exchange with c10d as the transport
(kept for hosts that already own a process group, and as bench.py's fall-back).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_frames(nframes, world, rank):
    """Contiguous block partition of a batch; returns (first, count) for `rank` (orbfe_mc_shard)."""
    from . import binding
    return binding.mc_shard(nframes, world, rank)


def owner_of_frame(f, nframes, world):
    base, rem = divmod(nframes, world)
    cut = rem * (base + 1)
    return f // (base + 1) if f < cut else rem + (f - cut) // max(base, 1)


def _work_done(w):
    """True only for a collective that has finished WITHOUT an error.  c10d's is_completed() is also true for a
    collective that ended with an exception (WorkNCCL::isCompleted = exception() || finishedGPUExecution), and
    wait() is the only call that raises it: a failed or aborted all-gather must never be skipped silently, or the
    pipeline would go on reading a stale gathered buffer."""
    try:
        if not bool(w.is_completed()):
            return False
        # c10d: is_success() is false for a collective that ended with an exception.  (Work.exception() cannot be called
        # from Python at all: pybind has no converter for std::exception_ptr and raises TypeError.)
        ok = getattr(w, "is_success", None)
        if ok is not None:
            return bool(ok())
        exc = getattr(w, "exception", None)  # test doubles
        return exc is None or exc() is None
    except Exception:  # a backend without the queries: keep the ordering wait (which also raises)
        return False


def ring_pairs(world, frames, rank, hops=(1,)):
    """Cross-camera pairs of a rig whose cameras form a ring (global frame g = rank * frames + local index):
    every local frame is a query against the frames `h` cameras further round the ring, for h in hops.  The last
    local frames' partners live on the next rank -- the reason for the all-gather."""
    from . import binding
    return binding.mc_ring_pairs(world, frames, rank, hops)  # orbfe_mc_ring_pairs


def job_offsets(frames, cap, slab_bytes, pairs):
    """Byte offsets of the job records of `pairs` = [(local query frame, global train frame)]: per job
    (query descriptors, query count) inside this rank's slab and (train descriptors, train count) inside the
    gathered buffer.  Shared by the device matcher and by its CPU stand-in in the tests."""
    from . import binding
    assert slab_bytes == binding.mc_layout(frames, cap)[2]
    return binding.mc_job_offsets(frames, cap, pairs)  # orbfe_mc_job_offsets


class CrossCameraMatcher:
    """knn-2 of this rank's query frames against their partner frames in the gathered buffer, one launch per batch,
    everything resident in HBM.  One job table per exchange buffer (the buffers are persistent, so the tables are
    built and uploaded once)."""

    def __init__(self, exchanges, pairs, device):
        from . import binding
        self.binding = binding
        self.pairs = list(pairs)
        self.njobs = len(self.pairs)
        x0 = exchanges[0]
        self.cap = x0.cap
        self.dev_index = device.index if device.index is not None else 0
        off = job_offsets(x0.frames, x0.cap, x0.slab_bytes, self.pairs)
        self.tables = {}
        for x in exchanges:
            rec = np.zeros(self.njobs, binding.KNN2_JOB_DTYPE)
            rec["q_desc"] = x.slab.data_ptr() + off[:, 0]
            rec["q_count"] = x.slab.data_ptr() + off[:, 1]
            rec["t_desc"] = x.gathered.data_ptr() + off[:, 2]
            rec["t_count"] = x.gathered.data_ptr() + off[:, 3]
            self.tables[id(x)] = torch.from_numpy(rec.view(np.uint8).copy()).to(device)
        self.idx = torch.full((max(self.njobs, 1), self.cap, 2), -1, dtype=torch.int32, device=device)
        self.dist = torch.full((max(self.njobs, 1), self.cap, 2), -1, dtype=torch.int32, device=device)

    def match(self, x, stream=None):
        """Queue the launch for exchange `x` (whose gathered buffer must be final on `stream`: wait for the
        collective's handle first) on `stream` (default: torch's current stream).  Returns (idx, dist) tensors of
        shape [njobs, cap, 2]; rows beyond a query frame's count keep their previous content."""
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        self.binding.bfknn2_frames_device(self.tables[id(x)].data_ptr(), self.njobs, self.cap, self.idx.data_ptr(),
                                          self.dist.data_ptr(), stream=st, device=self.dev_index)
        return self.idx, self.dist


class PipelinedExchange:
    """Two DescriptorExchange buffers used alternately: the all-gather of batch i runs while batch i+1 is being
    extracted into the other slab (collectives overlapped with compute on a separate stream).  Usage per batch:
        x = pipe.begin()          # waits for the collective that last used this slab pair, returns the exchange
        ... extractor writes into x.desc_view() / x.count_view() ...
        pipe.submit()             # starts this batch's all-gather asynchronously
    and `pipe.drain()` after the last batch; `pipe.completed()` is the most recent exchange whose gathered
    buffer is final after drain()."""

    def __init__(self, frames_per_rank, cap, device, world=None, rank=None, depth=2, producer_stream=None):
        """producer_stream: the torch stream the extractor writes the slabs on, when that is not the stream that is
        current when submit() is called (the collective orders itself after the CURRENT stream only; an extractor
        context runs on a private non-blocking stream unless set_stream() gave it one)."""
        self.x = [DescriptorExchange(frames_per_rank, cap, device, world, rank) for _ in range(depth)]
        self.pending = [None] * depth
        self.i = 0
        self.last = None
        self.producer_stream = producer_stream
        self.waits_skipped = 0  # collectives found cleanly completed in begin() (no cross-stream wait needed)

    def begin(self):
        k = self.i % len(self.x)
        w = self.pending[k]
        if w is not None:
            # a collective issued two batches ago has normally finished: then nothing has to be ordered, and the
            # cross-stream wait (a barrier packet plus a signal, ~8 us of queue bubble per batch on this stack) is
            # skipped; only a collective still in flight makes the current stream wait for it
            if _work_done(w):
                self.waits_skipped += 1
            else:
                w.wait()  # orders the current stream after the collective -- and raises if it failed
            self.pending[k] = None
        return self.x[k]

    def submit(self):
        k = self.i % len(self.x)
        self.pending[k] = self.x[k].all_gather_async(self.producer_stream)
        self.last = self.x[k]
        self.i += 1

    def drain(self):
        for k, w in enumerate(self.pending):
            if w is not None:
                w.wait()
                self.pending[k] = None

    def completed(self):
        return self.last


class DescriptorExchange:
    """Owns this rank's slab and the gathered buffer; all_gather() is one collective per batch."""

    def __init__(self, frames_per_rank, cap, device, world=None, rank=None):
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank() if dist.is_initialized() else 0)
        self.frames, self.cap = frames_per_rank, cap
        from . import binding
        self.desc_bytes, _, self.slab_bytes = binding.mc_layout(frames_per_rank, cap)  # orbfe_mc_layout
        self.slab = torch.zeros(self.slab_bytes, dtype=torch.uint8, device=device)
        self.gathered = torch.zeros(self.world * self.slab_bytes, dtype=torch.uint8, device=device)

    # views the extractor writes into
    def desc_view(self):
        return self.slab[: self.desc_bytes].view(self.frames, self.cap, 32)

    def count_view(self):
        return self.slab[self.desc_bytes: self.desc_bytes + 4 * self.frames].view(torch.int32)

    def all_gather(self):
        if self.world == 1:
            self.gathered.copy_(self.slab)
        else:
            dist.all_gather_into_tensor(self.gathered, self.slab)
        return self.gathered

    def unpack(self, r):
        """(counts[frames] int32, desc[frames, cap, 32]) contributed by rank r."""
        s = self.gathered[r * self.slab_bytes:(r + 1) * self.slab_bytes]
        return (s[self.desc_bytes: self.desc_bytes + 4 * self.frames].view(torch.int32),
                s[: self.desc_bytes].view(self.frames, self.cap, 32))

    def all_gather_async(self, producer_stream=None):
        """Start the collective and return its work handle (None when there is nothing to wait for).  With the
        "nccl" backend the collective runs on the process group's own stream, ordered after everything already
        queued on the CURRENT stream -- so the next batch's kernels overlap it; call `.wait()` on the handle
        (makes the current stream wait) before the slab or the gathered buffer is touched again.
        The slab must have been written on the current stream, or on `producer_stream` (a torch.cuda.Stream /
        ExternalStream), which the current stream is then made to wait for first: a collective that is not ordered
        after the extractor's K-DESC would race with its writes."""
        if producer_stream is not None and self.slab.is_cuda:
            cur = torch.cuda.current_stream(self.slab.device)
            if producer_stream.cuda_stream != cur.cuda_stream:
                cur.wait_stream(producer_stream)
        if self.world == 1 and not dist.is_initialized():
            self.gathered.copy_(self.slab)
            return None
        return dist.all_gather_into_tensor(self.gathered, self.slab, async_op=True)

    def query_shard(self):
        """Cross-camera matching is sharded by query frame: global frame ids this rank matches."""
        total = self.world * self.frames
        first, count = shard_frames(total, self.world, self.rank)
        return range(first, first + count)
