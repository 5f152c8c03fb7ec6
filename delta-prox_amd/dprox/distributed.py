"""Batch sharding over the GPUs of a node: one process per GPU.

Every operation of the solvers is per image, so the batch dimension is split into contiguous
slices, each rank solves its slice with its own compiled solver, and only three collectives exist, none of them inside an
iteration:

  * ``broadcast_constants`` -- shared constants built once on one rank (PSF / OTF / denominator tables, denoiser weights,
    sampling masks, rho / lambda schedules) reach the others over xGMI instead of being rebuilt on every rank;
  * ``scatter_batch``       -- a batch held by one rank is dealt out;
  * ``all_gather_batch``    -- the per-rank results are collected.

Transport: ``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests) or, with
``Comm`` below, RCCL through the library's own C ABI (``dpx_comm_*``: raw device pointers on the caller's stream, no
torch types -- what a non-Python host would bind).  The reference has no counterpart (single device,
dprox/algo/base.py:118).

Caveat kept from the reference: the CG stop rule couples the images of a batch (linalg/solve/solver_cg.py:103-104),
so CG-based solves are reproducible per shard, not across different shardings.

REQUIREMENT -- one process per GPU, one solve at a time per process.  ``solve()`` is not re-entrant (as the reference's is not:
module-level state in linop/comp_graph.py:201-202): the x-only flag of a solve (algo/driver.py), the sub-batch chains' share of
the GPU (``dpx_admm_iter_share``) and the tuning registry are per-process state.  Launch as ``bench.py`` is launched
(``torch.distributed.run --nproc-per-node N``: RANK / LOCAL_RANK from the environment, ``torch.cuda.set_device(LOCAL_RANK)``);
several solver threads inside one process, or one process driving several GPUs, are not supported.
"""
import ctypes
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import _backend as be


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


class Comm:
    """An RCCL communicator behind the C ABI (``dpx_comm_*``), bound to this process's current HIP device.
    ``Comm.from_process_group()`` ships the RCCL unique id through the (already initialised) torch.distributed group -- host
    side only; the data path then runs on raw device pointers.  ``Comm.single()`` is the one-rank communicator."""

    def __init__(self, handle, rank, world):
        self._h, self.rank, self.world = handle, rank, world

    @staticmethod
    def _unique_id():
        buf = ctypes.create_string_buffer(128)
        be.lib().call("dpx_comm_unique_id", buf)
        return buf.raw

    @classmethod
    def _init(cls, uid, rank, world):
        h = ctypes.c_void_p()
        be.lib().call("dpx_comm_init", ctypes.byref(h), ctypes.create_string_buffer(uid, 128), rank, world)
        return cls(h, rank, world)

    @classmethod
    def single(cls):
        return cls._init(cls._unique_id(), 0, 1)

    @classmethod
    def from_process_group(cls, group=None):
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        uid = _bcast_meta(cls._unique_id() if rank == 0 else None, 0, group)
        return cls._init(uid, rank, world)

    def broadcast(self, t: torch.Tensor, root=0):
        be.require(t, dtype=None, what="broadcast buffer")
        be.lib().call("dpx_comm_broadcast", self._h, be.ptr(t), t.numel() * t.element_size(), root, be.stream())
        return t

    def all_gather(self, local: torch.Tensor):
        """[n, ...] per rank (same n everywhere) -> [world * n, ...]"""
        be.require(local, dtype=None, what="all_gather input")
        out = torch.empty((self.world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        be.lib().call("dpx_comm_allgather", self._h, be.ptr(local), be.ptr(out), local.numel() * local.element_size(), be.stream())
        return out

    def scatter(self, full: Optional[torch.Tensor], n_per_rank: int, tail_shape, dtype, device, root=0):
        """root holds [world * n_per_rank, ...]; every rank receives [n_per_rank, ...]"""
        recv = torch.empty((n_per_rank,) + tuple(tail_shape), dtype=dtype, device=device)
        be.lib().call("dpx_comm_scatter", self._h, be.ptr(full) if self.rank == root else None, be.ptr(recv),
                      recv.numel() * recv.element_size(), root, be.stream())
        return recv

    def close(self):
        if self._h is not None:
            be.lib().call("dpx_comm_destroy", self._h)
            self._h = None


def broadcast_constants(tensors: Optional[Dict[str, torch.Tensor]], src: int = 0, group=None, device=None, comm: Optional[Comm] = None):
    """One-time broadcast of shared constants.  Rank `src` passes ``{name: tensor}``; every rank gets the same dict back (on
    `device`).  Shapes / dtypes travel as a small object, the payloads as one RCCL (or gloo) broadcast each."""
    rank = dist.get_rank(group)
    meta = _bcast_meta({k: (tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in tensors.items()} if rank == src else None, src, group)
    out = {}
    for k in sorted(meta):
        shape, dtype = meta[k][0], getattr(torch, meta[k][1])
        if rank == src:
            t = tensors[k].to(device if device is not None else tensors[k].device).contiguous()
        else:
            t = torch.empty(shape, dtype=dtype, device=device if device is not None else "cpu")
        if comm is not None and (t.is_cuda or be.host_mode()):
            comm.broadcast(t, src)
        else:
            dist.broadcast(t, src=src, group=group)
        out[k] = t
    return out


def export_tables(solver, shape, device) -> Dict[str, torch.Tensor]:
    """The per-shape constants of a compiled solver whose x-update is a Fourier division -- OTF tables of the data terms'
    convolutions and the two |OTF|^2 sums of the denominator (opaque half-spectrum layouts) -- built if they are not cached yet.
    They depend on (C, H, W) only, so the rank that holds them can broadcast them (``share_tables``) instead of every rank
    evaluating its own copies (dpx_psf2otf: a direct fp64 DFT per table)."""
    from .linop.fourier import conv
    from .algo.fused import _omega_conv
    ls = solver.least_square
    out = {}
    for i, fn in enumerate(solver.omega_fns):
        cv = _omega_conv(fn)
        if type(cv) is conv:
            out[f"otf{i}"] = cv._tables(shape, device)
    (t0, c0), (t1, c1) = ls.diag_tables(shape, device, True)
    for name, t in (("t0", t0), ("t1", t1)):
        if isinstance(t, torch.Tensor):
            out[name] = t
    out["consts"] = torch.tensor([c0, c1], dtype=torch.float64, device=device)
    return out


def import_tables(solver, shape, device, tabs: Dict[str, torch.Tensor]):
    """fills the solver's caches with tables received from another rank (same problem, same plane size)"""
    from .linop.fourier import conv
    from .algo.fused import _omega_conv
    ls = solver.least_square
    for i, fn in enumerate(solver.omega_fns):
        cv = _omega_conv(fn)
        if type(cv) is conv and f"otf{i}" in tabs:
            cv.cache[(tuple(shape[1:]), str(device))] = tabs[f"otf{i}"]
    c0, c1 = (float(v) for v in tabs["consts"].cpu())
    key = (tuple(shape[1:]), str(device), True) + tuple(fn.linop.tables_version() for fn in list(ls.quad_fns) + list(ls.other_fns))
    ls._diag_cache = (key, ((tabs.get("t0"), c0), (tabs.get("t1"), c1)))


def share_tables(solver, shape, src: int = 0, group=None, device=None, comm: Optional[Comm] = None):
    """rank `src` builds the solver's tables for `shape` (any batch size) and broadcasts them; every rank's solver ends up with
    them cached.  One-time, outside any timed solve."""
    rank = dist.get_rank(group)
    device = device if device is not None else solver.device
    tabs = broadcast_constants(export_tables(solver, shape, device) if rank == src else None, src, group, device, comm)
    if rank != src:
        import_tables(solver, shape, device, tabs)
    return tabs


def scatter_batch(full: Optional[torch.Tensor], src: int = 0, group=None, device=None, comm: Optional[Comm] = None) -> torch.Tensor:
    """rank `src` holds the full [B, ...] tensor; every rank receives its slice (possibly empty).  An evenly divisible batch is
    dealt out from views of `full` (no staging copies); a ragged one is padded to the longest slice."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    meta = _bcast_meta((tuple(full.shape), str(full.dtype).replace("torch.", "")) if rank == src else None, src, group)
    shape, dtype = meta[0], getattr(torch, meta[1])
    sl = shard_slices(shape[0], world)
    device = device if device is not None else (full.device if full is not None else torch.device("cpu"))
    nmax = max(b - a for a, b in sl)
    even = all(b - a == nmax for a, b in sl)
    if rank == src:
        full = full.to(device).contiguous()
        if not even:                                      # pad the short slices: [world, nmax, ...]
            padded = torch.zeros((world * nmax,) + tuple(shape[1:]), dtype=dtype, device=device)
            for r, (a, b) in enumerate(sl):
                padded[r * nmax:r * nmax + (b - a)] = full[a:b]
            full = padded
    if comm is not None and (torch.device(device).type == "cuda" or be.host_mode()):
        recv = comm.scatter(full if rank == src else None, nmax, shape[1:], dtype, device, src)
    else:
        recv = torch.empty((nmax,) + tuple(shape[1:]), dtype=dtype, device=device)
        dist.scatter(recv, scatter_list=[full[r * nmax:(r + 1) * nmax] for r in range(world)] if rank == src else None, src=src, group=group)
    a, b = sl[rank]
    return recv[:b - a] if b - a == nmax else recv[:b - a].contiguous()


def all_gather_batch(local: torch.Tensor, batch: int, group=None, comm: Optional[Comm] = None) -> torch.Tensor:
    """inverse of the sharding: every rank ends up with the full [batch, ...] tensor"""
    world = dist.get_world_size(group)
    sl = shard_slices(batch, world)
    nmax = max(b - a for a, b in sl)
    even = all(b - a == nmax for a, b in sl)
    if even:
        send = local.contiguous()
    else:
        send = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[:local.shape[0]] = local
    if comm is not None and (send.is_cuda or be.host_mode()):
        flat = comm.all_gather(send)
    else:
        flat = torch.empty((world * nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(flat, send, group=group)
    if even:
        return flat
    return torch.cat([flat[r * nmax:r * nmax + (b - a)] for r, (a, b) in enumerate(sl)], dim=0)


def solve_sharded(local_solve: Callable[[Dict[str, torch.Tensor]], torch.Tensor], inputs: Optional[Dict[str, torch.Tensor]],
                  src: int = 0, group=None, device=None, comm: Optional[Comm] = None) -> torch.Tensor:
    """Scatter every batched tensor of `inputs` (held by rank `src`), run `local_solve` on the local slice, all-gather.

    `local_solve` receives a dict with the same keys and must return a [b_local, ...] tensor (it is not called on ranks
    whose slice is empty)."""
    rank = dist.get_rank(group)
    keys = _bcast_meta(sorted(inputs) if rank == src else None, src, group)
    local = {k: scatter_batch(inputs[k] if rank == src else None, src, group, device, comm) for k in keys}
    batch = _bcast_meta(int(inputs[keys[0]].shape[0]) if rank == src else None, src, group)
    n_local = local[keys[0]].shape[0]
    out = local_solve(local) if n_local > 0 else None
    world = dist.get_world_size(group)
    if batch % world == 0:                                # every rank has work: the output shape is known locally
        return all_gather_batch(out, batch, group, comm)
    # ranks with an empty slice still take part in the collective: learn the output shape from a neighbour
    meta = (tuple(out.shape[1:]), str(out.dtype).replace("torch.", "")) if out is not None else None
    metas = [None] * world
    dist.all_gather_object(metas, meta, group=group)
    shape, dtype = next(m for m in metas if m is not None)
    if out is None:
        out = torch.empty((0,) + tuple(shape), dtype=getattr(torch, dtype), device=local[keys[0]].device)
    return all_gather_batch(out, batch, group, comm)
