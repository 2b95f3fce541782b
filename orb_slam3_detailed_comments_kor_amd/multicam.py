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

    def query_shard(self):
        """Cross-camera matching is sharded by query frame: global frame ids this rank matches."""
        total = self.world * self.frames
        first, count = shard_frames(total, self.world, self.rank)
        return range(first, first + count)
