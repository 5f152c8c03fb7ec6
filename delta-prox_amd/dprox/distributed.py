"""Batch sharding over the GPUs of a node: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" in the CPU tests).

Every operation of the solvers is per image, so the batch dimension is split into contiguous
slices, each rank solves its slice with its own compiled solver, and only the inputs (scatter) and the result
(all-gather) cross the links -- there is no per-iteration communication and no collective on the data path.
The reference has no counterpart (single device, dprox/algo/base.py:118).

Caveat kept from the reference: the CG stop rule couples the images of a batch (linalg/solve/solver_cg.py:103-104),
so CG-based solves are reproducible per shard, not across different shardings.
"""
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_slices(batch: int, world: int) -> List[Tuple[int, int]]:
    """contiguous, near-equal slices; trailing ranks may be empty when batch < world"""
    base, extra = divmod(batch, world)
    out, start = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((start, start + n))
        start += n
    return out


def _bcast_meta(obj, src, group):
    box = [obj]
    dist.broadcast_object_list(box, src=src, group=group)
    return box[0]


def scatter_batch(full: Optional[torch.Tensor], src: int = 0, group=None, device=None) -> torch.Tensor:
    """rank `src` holds the full [B, ...] tensor; every rank receives its slice (possibly empty)"""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    meta = _bcast_meta((tuple(full.shape), str(full.dtype).replace("torch.", "")) if rank == src else None, src, group)
    shape, dtype = meta[0], getattr(torch, meta[1])
    sl = shard_slices(shape[0], world)
    device = device if device is not None else (full.device if full is not None else torch.device("cpu"))
    nmax = max(b - a for a, b in sl)
    recv = torch.empty((nmax,) + tuple(shape[1:]), dtype=dtype, device=device)
    if rank == src:
        chunks = []
        for a, b in sl:                                  # equal-size chunks for dist.scatter (padded)
            c = torch.zeros((nmax,) + tuple(shape[1:]), dtype=dtype, device=device)
            c[:b - a] = full[a:b].to(device)
            chunks.append(c)
        dist.scatter(recv, scatter_list=chunks, src=src, group=group)
    else:
        dist.scatter(recv, scatter_list=None, src=src, group=group)
    a, b = sl[rank]
    return recv[:b - a].contiguous()


def all_gather_batch(local: torch.Tensor, batch: int, group=None) -> torch.Tensor:
    """inverse of the sharding: every rank ends up with the full [batch, ...] tensor"""
    world = dist.get_world_size(group)
    sl = shard_slices(batch, world)
    nmax = max(b - a for a, b in sl)
    pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:b - a] for p, (a, b) in zip(parts, sl)], dim=0)


def solve_sharded(local_solve: Callable[[Dict[str, torch.Tensor]], torch.Tensor], inputs: Optional[Dict[str, torch.Tensor]],
                  src: int = 0, group=None, device=None) -> torch.Tensor:
    """Scatter every batched tensor of `inputs` (held by rank `src`), run `local_solve` on the local slice, all-gather.

    `local_solve` receives a dict with the same keys and must return a [b_local, ...] tensor (it is not called on ranks
    whose slice is empty)."""
    rank = dist.get_rank(group)
    keys = _bcast_meta(sorted(inputs) if rank == src else None, src, group)
    local = {k: scatter_batch(inputs[k] if rank == src else None, src, group, device) for k in keys}
    batch = _bcast_meta(int(inputs[keys[0]].shape[0]) if rank == src else None, src, group)
    n_local = local[keys[0]].shape[0]
    out = local_solve(local) if n_local > 0 else None
    # ranks with an empty slice still take part in the collective: learn the output shape from a neighbour
    meta = (tuple(out.shape[1:]), str(out.dtype).replace("torch.", "")) if out is not None else None
    metas = [None] * dist.get_world_size(group)
    dist.all_gather_object(metas, meta, group=group)
    shape, dtype = next(m for m in metas if m is not None)
    if out is None:
        out = torch.empty((0,) + tuple(shape), dtype=getattr(torch, dtype), device=local[keys[0]].device)
    return all_gather_batch(out, batch, group)
