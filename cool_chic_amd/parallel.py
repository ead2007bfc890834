"""Frame-level sharding over the GPUs of one node.

Every cool-chic decodes from its own byte ranges, so frames are independent units: frame i goes to
rank i mod world_size, each rank decodes its frames in one DecodeBatch, and the only communication is
one gather of the decoded integer planes to the writer rank (RCCL over xGMI when the backend is
"nccl"; "gloo" in the CPU tests). No collective runs inside a frame.  A video GOP adds the path's one real
exchange: a decoded frame travels point to point to the ranks whose frames predict from it.

`dst` / `src` arguments are ranks of `group` (translated to global ranks for torch.distributed)."""
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
    dist.gather(padded, bucket, dst=_global_rank(group, dst), group=group)
    if rank != dst:
        return None
    return [b[: int(s.item())] for b, s in zip(bucket, sizes)]


class EqualSizeGather:
    """Gather of same-sized byte messages to `dst` with persistent buffers: no size exchange, no host synchronisation, no
    per-call allocation (the steady-state path when every rank decodes frames of the same geometry).  With the "gloo"
    backend (CPU tests, ranks sharing one GPU) the message is staged through pinned host memory: gloo gathers CPU tensors."""

    def __init__(self, n_bytes: int, device, dst: int = 0, group=None):
        self.dst, self.group = dst, group
        self.local = torch.empty(n_bytes, dtype=torch.uint8, device=device)
        self.staged = dist.get_backend(group) == "gloo" and self.local.is_cuda
        wire = torch.empty(n_bytes, dtype=torch.uint8, pin_memory=True) if self.staged else self.local
        self.wire = wire
        world = dist.get_world_size(group)
        self.bucket = [torch.empty_like(wire) for _ in range(world)] if dist.get_rank(group) == dst else None

    def __call__(self, planes: Sequence[torch.Tensor]) -> Optional[List[torch.Tensor]]:
        torch.cat([p.contiguous().view(torch.uint8).reshape(-1) for p in planes], out=self.local)
        if self.staged:
            self.wire.copy_(self.local)  # synchronous device -> host copy on the current stream
        dist.gather(self.wire, self.bucket, dst=_global_rank(self.group, self.dst), group=self.group)
        return self.bucket


def unshard(per_rank_items: Sequence[Sequence], n_frames: int) -> list:
    """Inverse of shard_indices: per_rank_items[r][k] is frame r + k * world_size."""
    world = len(per_rank_items)
    out = [None] * n_frames
    for r, items in enumerate(per_rank_items):
        for k, it in enumerate(items):
            out[r + k * world] = it
    return out


# ---- a GOP sharded over ranks (video) --------------------------------------------------------
def gop_owner(coding_index: int, world_size: int) -> int:
    """Frame at coding index k is decoded and reconstructed by rank k mod world_size."""
    return coding_index % world_size


def _global_rank(group, r: int) -> int:
    """Collectives / point-to-point calls take GLOBAL ranks; `r` is a rank of `group`."""
    return r if group is None else dist.get_global_rank(group, r)


def send_planes(planes: Sequence[torch.Tensor], dst: int, device, group=None) -> None:
    """The owner's integer planes of one frame -> rank `dst` of `group`: one packed message, point to point (RCCL
    send / recv over the direct xGMI link with the "nccl" backend; staged through the host with "gloo")."""
    staged = dist.get_backend(group) == "gloo"
    buf = pack_planes([p.to("cpu" if staged else device) for p in planes])
    dist.send(buf, dst=_global_rank(group, dst), group=group)


def recv_planes(specs: Sequence, src: int, device, group=None) -> List[torch.Tensor]:
    """Counterpart of send_planes; `specs` = [(shape, dtype), ...] comes from the frame header, so no sizes travel."""
    n_bytes = [int(torch.Size(s).numel()) * torch.empty(0, dtype=d).element_size() for s, d in specs]
    staged = dist.get_backend(group) == "gloo"
    buf = torch.empty(sum(n_bytes), dtype=torch.uint8, device="cpu" if staged else device)
    dist.recv(buf, src=_global_rank(group, src), group=group)
    out, off = [], 0
    for (shape, dtype), nb in zip(specs, n_bytes):
        out.append(buf[off:off + nb].view(dtype).reshape(shape).to(device))
        off += nb
    return out


def gop_consumers(n_frames: int, references, world_size: int, collect: Optional[int] = 0) -> List[List[int]]:
    """For every frame (coding index) the ranks that need its planes besides its owner: the owners of the frames that
    predict from it (at most two per frame in a hierarchical GOP, decode.py:156-189) and the rank that collects the
    sequence (None: nobody)."""
    out = []
    for k in range(n_frames):
        need = {gop_owner(j, world_size) for j in range(k + 1, n_frames) if k in references[j]}
        if collect is not None:
            need.add(collect)
        need.discard(gop_owner(k, world_size))
        out.append(sorted(need))
    return out


def run_sharded_gop(n_frames: int, plane_specs, references, produce, device="cpu", group=None, collect: Optional[int] = 0) -> dict:
    """Coding-order schedule of a GOP whose frames are spread round-robin over the ranks.

    plane_specs[k]  [(shape, dtype) x 3] of frame k (coding index), known to every rank
    references[k]   coding indices of the frames frame k predicts from (all < k)
    produce(k, refs) -> planes of frame k; called ONLY on gop_owner(k); refs = planes of references[k]
    collect         rank that ends up with every frame (the writer), or None
    Returns {k: planes} holding, on each rank, the frames it produced or received: everything on `collect`.

    The expensive part of `produce` (the cool-chic decodes) does not depend on the references, so an owner runs it for
    all of its frames up front and `produce` only reconstructs.  A frame's planes travel point to point, only to the
    ranks whose frames predict from it (and to the collector): the one real exchange step of the path."""
    initialised = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if initialised else 1
    rank = dist.get_rank(group) if initialised else 0
    consumers = gop_consumers(n_frames, references, world, collect if world > 1 else None)
    done = {}
    for k in range(n_frames):
        owner = gop_owner(k, world)
        if rank == owner:
            done[k] = list(produce(k, [done[r] for r in references[k]]))
            for dst in consumers[k]:
                send_planes(done[k], dst, device, group)
        elif rank in consumers[k]:
            done[k] = recv_planes(plane_specs[k], owner, device, group)
    return done
