"""Multi-camera / multi-frame sharding across the GPUs of one node.

Independent image pyramids shard naturally: frame f of a batch goes to one rank, each rank runs
the single-GPU batched extractor on its shard, and ONE all-gather of fixed-size descriptor slabs
(RCCL over xGMI when the backend is "nccl") gives every rank all descriptors for cross-camera
matching.  There is no reduction anywhere on this path.

Slab layout per rank (one contiguous uint8 tensor, so the exchange is a single collective):
    [ frames_per_rank * cap * 32 bytes of descriptors | frames_per_rank * int32 keypoint counts ]
The extractor writes straight into views of the slab, so no packing copy is needed.
"""
import torch
import torch.distributed as dist


def shard_frames(nframes, world, rank):
    """Contiguous block partition of a batch; returns (first, count) for `rank`."""
    base, rem = divmod(nframes, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def owner_of_frame(f, nframes, world):
    base, rem = divmod(nframes, world)
    cut = rem * (base + 1)
    return f // (base + 1) if f < cut else rem + (f - cut) // max(base, 1)


def _work_done(w):
    try:
        return bool(w.is_completed())
    except Exception:  # a backend without the query: keep the ordering wait
        return False


class PipelinedExchange:
    """Two DescriptorExchange buffers used alternately: the all-gather of batch i runs while batch i+1 is being
    extracted into the other slab (collectives overlapped with compute on a separate stream).  Usage per batch:
        x = pipe.begin()          # waits for the collective that last used this slab pair, returns the exchange
        ... extractor writes into x.desc_view() / x.count_view() ...
        pipe.submit()             # starts this batch's all-gather asynchronously
    and `pipe.drain()` after the last batch; `pipe.completed()` is the most recent exchange whose gathered
    buffer is final after drain()."""

    def __init__(self, frames_per_rank, cap, device, world=None, rank=None, depth=2):
        self.x = [DescriptorExchange(frames_per_rank, cap, device, world, rank) for _ in range(depth)]
        self.pending = [None] * depth
        self.i = 0
        self.last = None

    def begin(self):
        k = self.i % len(self.x)
        w = self.pending[k]
        if w is not None:
            # a collective issued two batches ago has normally finished: then nothing has to be ordered, and the
            # cross-stream wait (a barrier packet plus a signal, ~8 us of queue bubble per batch on this stack) is
            # skipped; only a collective still in flight makes the current stream wait for it
            if not _work_done(w):
                w.wait()
            self.pending[k] = None
        return self.x[k]

    def submit(self):
        k = self.i % len(self.x)
        self.pending[k] = self.x[k].all_gather_async()
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
        self.desc_bytes = frames_per_rank * cap * 32
        self.slab_bytes = self.desc_bytes + frames_per_rank * 4
        self.slab = torch.zeros(self.slab_bytes, dtype=torch.uint8, device=device)
        self.gathered = torch.zeros(self.world * self.slab_bytes, dtype=torch.uint8, device=device)

    # views the extractor writes into
    def desc_view(self):
        return self.slab[: self.desc_bytes].view(self.frames, self.cap, 32)

    def count_view(self):
        return self.slab[self.desc_bytes:].view(torch.int32)

    def all_gather(self):
        if self.world == 1:
            self.gathered.copy_(self.slab)
        else:
            dist.all_gather_into_tensor(self.gathered, self.slab)
        return self.gathered

    def unpack(self, r):
        """(counts[frames] int32, desc[frames, cap, 32]) contributed by rank r."""
        s = self.gathered[r * self.slab_bytes:(r + 1) * self.slab_bytes]
        return s[self.desc_bytes:].view(torch.int32), s[: self.desc_bytes].view(self.frames, self.cap, 32)

    def all_gather_async(self):
        """Start the collective and return its work handle (None when there is nothing to wait for).  With the
        "nccl" backend the collective runs on the process group's own stream, ordered after everything already
        queued on the current stream -- so the next batch's kernels overlap it; call `.wait()` on the handle
        (makes the current stream wait) before the slab or the gathered buffer is touched again."""
        if self.world == 1 and not dist.is_initialized():
            self.gathered.copy_(self.slab)
            return None
        return dist.all_gather_into_tensor(self.gathered, self.slab, async_op=True)

    def query_shard(self):
        """Cross-camera matching is sharded by query frame: global frame ids this rank matches."""
        total = self.world * self.frames
        first, count = shard_frames(total, self.world, self.rank)
        return range(first, first + count)
