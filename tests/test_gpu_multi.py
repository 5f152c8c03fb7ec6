"""-m gpu: the RCCL side of the batch sharding.  One-rank communicator on any box (the C-ABI wrappers load librccl, create a
communicator on this GPU and run the three collectives); the two-rank test spawns one process per GPU and skips itself on
boxes with a single device."""
import os
import socket

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_c_abi_communicator_single_rank():
    from dprox import distributed as dd
    c = dd.Comm.single()
    assert (c.rank, c.world) == (0, 1)
    t = torch.arange(1000, dtype=torch.float32, device="cuda")
    c.broadcast(t, 0)
    g = c.all_gather(t.view(10, 100))
    s = c.scatter(t.view(10, 100), 10, (100,), torch.float32, t.device, 0)
    torch.cuda.synchronize()
    assert torch.equal(g, t.view(10, 100)) and torch.equal(s, t.view(10, 100))
    z = torch.randn(3, 5, dtype=torch.complex64, device="cuda")
    assert torch.equal(c.all_gather(z), z)
    c.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, use_abi):
    import sys
    for p in (ROOT, os.path.join(ROOT, "delta-prox_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import synthetic
    import dprox as dp
    from dprox import distributed as dd
    comm = dd.Comm.from_process_group() if use_abi else None
    B = 5                                                 # ragged: 3 + 2
    gt, b, psf0 = synthetic.deconv_case(B, 3, 64, 64, seed=21, ksize=7, ksigma=2.0)
    consts = dd.broadcast_constants({"psf": torch.from_numpy(psf0).to(dev)} if rank == 0 else None, src=0, device=dev, comm=comm)
    psf = consts["psf"].cpu().numpy()

    def local_solve(loc):
        bb = loc["b"]
        x = dp.Variable()
        fns = dp.sum_squares(dp.conv(x, psf) - bb) + dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1))
        return dp.compile(fns, method="admm", device=dev).solve(x0=bb, rhos=0.2, lams=0.01, max_iter=6)

    out = dd.solve_sharded(local_solve, {"b": torch.from_numpy(b).to(dev)} if rank == 0 else None, src=0, device=dev, comm=comm)
    if rank == 0:
        full = local_solve({"b": torch.from_numpy(b).to(dev)})
        q.put({"err": float((out - full).abs().max()), "shape": tuple(out.shape)})
    else:
        q.put(tuple(out.shape))
    dist.barrier()
    if comm is not None:
        comm.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("use_abi", [False, True])
def test_two_rank_sharded_solve_over_rccl(use_abi):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the round driver's multi-GPU node); single-GPU boxes run the one-rank communicator test")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, use_abi)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=300) for _ in procs]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    r0 = next(r for r in res if isinstance(r, dict))
    assert r0["shape"] == (5, 3, 64, 64) and r0["err"] == 0.0
