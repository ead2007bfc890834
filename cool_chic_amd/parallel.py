"""Frame-level sharding over the GPUs of one node.

Every cool-chic decodes from its own byte ranges, so frames are independent units: frame i goes to
rank i mod world_size, each rank decodes its frames in one DecodeBatch, and the only communication is
one gather of the decoded integer planes to the writer rank (RCCL over xGMI when the backend is
"nccl"; "gloo" in the CPU tests). No collective runs inside a frame."""
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_indices(n_frames: int, rank: int, world_size: int) -> List[int]:
    """Round-robin ownership: frame i -> rank i mod world_size."""
    return list(range(rank, n_frames, world_size))


def pack_planes(planes: Sequence[torch.Tensor]) -> torch.Tensor:
    """Flatten a list of uint8 / uint16 planes into one uint8 byte tensor (one message per rank)."""
    if not planes:
        return torch.empty(0, dtype=torch.uint8)
    return torch.cat([p.contiguous().view(torch.uint8).reshape(-1) for p in planes])


def gather_bytes(local: torch.Tensor, dst: int = 0, group=None) -> Optional[List[torch.Tensor]]:
    """Gather variable-length uint8 tensors to `dst` (sizes first, then one padded gather).
    Returns the list of per-rank tensors on `dst`, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    size = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size, group=group)
    n_max = int(max(int(s.item()) for s in sizes))
    padded = torch.zeros(n_max, dtype=torch.uint8, device=local.device)
    padded[: local.numel()] = local
    bucket = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bucket, dst=dst, group=group)
    if rank != dst:
        return None
    return [b[: int(s.item())] for b, s in zip(bucket, sizes)]


class EqualSizeGather:
    """Gather of same-sized byte messages to `dst` with persistent buffers: no size exchange, no host synchronisation, no
    per-call allocation (the steady-state path when every rank decodes frames of the same geometry)."""

    def __init__(self, n_bytes: int, device, dst: int = 0, group=None):
        self.dst, self.group = dst, group
        self.local = torch.empty(n_bytes, dtype=torch.uint8, device=device)
        world = dist.get_world_size(group)
        self.bucket = [torch.empty_like(self.local) for _ in range(world)] if dist.get_rank(group) == dst else None

    def __call__(self, planes: Sequence[torch.Tensor]) -> Optional[List[torch.Tensor]]:
        torch.cat([p.contiguous().view(torch.uint8).reshape(-1) for p in planes], out=self.local)
        dist.gather(self.local, self.bucket, dst=self.dst, group=self.group)
        return self.bucket


def unshard(per_rank_items: Sequence[Sequence], n_frames: int) -> list:
    """Inverse of shard_indices: per_rank_items[r][k] is frame r + k * world_size."""
    world = len(per_rank_items)
    out = [None] * n_frames
    for r, items in enumerate(per_rank_items):
        for k, it in enumerate(items):
            out[r + k * world] = it
    return out
