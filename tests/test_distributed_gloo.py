"""CPU, world_size 2, gloo: the N>1 path (batch sharding, scatter / all-gather, ragged and empty shards).
The per-rank solve is the CPU oracle here (test infrastructure); on the GPU box the same plumbing wraps the HIP solver."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, q):
    import sys
    for p in (ROOT, os.path.join(ROOT, "delta-prox_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as O
    import synthetic
    from dprox import distributed as dd

    psf = synthetic.point_spread_function(7, 2.0)
    gt, b, _ = synthetic.deconv_case(B, 1, 24, 32, seed=3, ksize=7, ksigma=2.0)

    def local_solve(loc):
        bb = loc["b"]
        terms = [O.sum_squares(O.lin_conv(psf).minus(bb)), O.norm1(O.lin_grad(0)), O.norm1(O.lin_grad(1))]
        return O.solve(terms, "admm", x0=bb, rhos=0.2, lams=0.01, max_iter=4)

    inputs = {"b": torch.from_numpy(b)} if rank == 0 else None
    out = dd.solve_sharded(local_solve, inputs, src=0)
    if rank == 0:
        full = local_solve({"b": torch.from_numpy(b)})
        q.put((float((out - full).abs().max()), tuple(out.shape), dd.shard_slices(B, world)))
    else:
        q.put(tuple(out.shape))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [5, 2, 1])
def test_sharded_solve_matches_single_process(B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0 = next(r for r in res if len(r) == 3)
    err, shape, slices = r0
    assert shape == (B, 1, 24, 32) and all(r == shape or (len(r) == 3) for r in res)
    assert err == 0.0, err                       # images never interact in the direct (Fourier) path: sharding is exact
    assert slices[0][0] == 0 and slices[-1][1] == B


def test_shard_slices():
    from dprox.distributed import shard_slices
    assert shard_slices(8, 8) == [(i, i + 1) for i in range(8)]
    assert shard_slices(8, 4) == [(0, 2), (2, 4), (4, 6), (6, 8)]
    assert shard_slices(5, 2) == [(0, 3), (3, 5)]
    assert shard_slices(1, 2) == [(0, 1), (1, 1)]
    assert shard_slices(32, 8)[-1] == (28, 32)
