"""CPU, world_size 2, gloo: the N>1 path (batch sharding, constant broadcast, scatter / all-gather, ragged and empty shards).
The per-rank solve is the PRODUCT solver -- the drop-in dprox API on the host-emulated build of the HIP kernels (tests/emul) --
so what gets sharded here is what gets sharded on the GPUs; the oracle only checks the gathered result."""
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
    for p in (ROOT, os.path.join(ROOT, "delta-prox_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import emul_util
    emul_util.use_emulator()                              # the unchanged kernel sources, compiled for the host
    import oracle as O
    import synthetic
    import dprox as dp
    from dprox import distributed as dd

    gt, b, psf0 = synthetic.deconv_case(B, 1, 24, 32, seed=3, ksize=7, ksigma=2.0)
    # shared constants live on rank 0 only and reach the others by broadcast (PSF, schedules)
    consts = dd.broadcast_constants({"psf": torch.from_numpy(psf0), "rhos": torch.full((4,), 0.2), "lams": torch.full((4,), 0.01)}
                                    if rank == 0 else None, src=0)
    psf = consts["psf"].numpy()

    def local_solve(loc):
        bb = loc["b"]
        x = dp.Variable()
        fns = dp.sum_squares(dp.conv(x, psf) - bb) + dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1))
        s = dp.compile(fns, method="admm", device="cpu")
        out = s.solve(x0=bb, rhos=consts["rhos"], lams=float(consts["lams"][0]), max_iter=4)
        assert s.last_path == "fused"
        return out

    inputs = {"b": torch.from_numpy(b)} if rank == 0 else None
    out = dd.solve_sharded(local_solve, inputs, src=0)
    if rank == 0:
        full = local_solve({"b": torch.from_numpy(b)})
        bt = torch.from_numpy(b)
        ref = O.solve([O.sum_squares(O.lin_conv(psf0).minus(bt)), O.norm1(O.lin_grad(0)), O.norm1(O.lin_grad(1))], "admm", x0=bt,
                      rhos=0.2, lams=0.01, max_iter=4)
        q.put({"err": float((out - full).abs().max()), "shape": tuple(out.shape), "slices": dd.shard_slices(B, world),
               "rel": float((out - ref).norm() / ref.norm())})
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
    try:
        res = [q.get(timeout=240) for _ in procs]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    r0 = next(r for r in res if isinstance(r, dict))
    err, shape, slices, rel = r0["err"], r0["shape"], r0["slices"], r0["rel"]
    assert shape == (B, 1, 24, 32) and all(isinstance(r, dict) or r == shape for r in res)
    assert err == 0.0, err                       # images never interact in the direct (Fourier) path: sharding is exact
    assert rel <= 1e-5, rel                      # ... and the gathered result is the reference's (oracle) answer
    assert slices[0][0] == 0 and slices[-1][1] == B


def test_shard_slices():
    from dprox.distributed import shard_slices
    assert shard_slices(8, 8) == [(i, i + 1) for i in range(8)]
    assert shard_slices(8, 4) == [(0, 2), (2, 4), (4, 6), (6, 8)]
    assert shard_slices(5, 2) == [(0, 3), (3, 5)]
    assert shard_slices(1, 2) == [(0, 1), (1, 1)]
    assert shard_slices(32, 8)[-1] == (28, 32)
