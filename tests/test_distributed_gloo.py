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


def _worker_c4(rank, world, port, B, q):
    """config 4's path per shard: masked-Fourier data term + nonneg + gray FFDNet prior (a small seeded network: the 15-layer one
    takes the emulator minutes), LinearizedADMM with the CG x-update.  The CG stop rule couples the images of a batch (SURVEY 8(e)
    caveat), so every shard is checked against an ORACLE run on the same sub-batch, not against a single-process run of the whole
    batch.  Also: the Fourier-path solver's tables built on rank 0 and shared (share_tables) give the same answer as local tables."""
    import sys
    for p in (ROOT, os.path.join(ROOT, "delta-prox_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import emul_util
    emul_util.use_emulator()
    import oracle as O
    import synthetic
    import dprox as dp
    from dprox import distributed as dd
    from dprox.contrib import masked_fft
    from dprox.linalg import LinearSolveConfig
    from dprox.proxfn.pnp.denoisers import FFDNet, FFDNetDenoiser
    from dprox.utils import ifft2

    H = W = 24
    gt, mask0, y0 = synthetic.csmri_case(B, H, W, seed=5, center=8)
    wts = synthetic.ffdnet_weights(13, 1, 1, 16, 3)
    # constants live on rank 0: sampling mask and denoiser weights reach the others by broadcast
    den = FFDNetDenoiser()
    den.model = FFDNet(in_nc=1, out_nc=1, nc=16, nb=3, act_mode="R")
    if rank == 0:
        den.model.load_layers(wts)
    consts = dd.broadcast_constants({"mask": torch.from_numpy(mask0)} if rank == 0 else None, src=0)
    sd = dd.broadcast_constants({k: v.detach() for k, v in den.state_dict().items()} if rank == 0 else None, src=0)
    if rank != 0:
        den.load_state_dict(sd, strict=True)
    yph, xv = dp.Placeholder(), dp.Variable()
    fns = dp.sum_squares(masked_fft(xv, consts["mask"]), yph) + dp.nonneg(xv) + dp.deep_prior(xv, denoiser=den)
    cfg = LinearSolveConfig(rtol=1e-6, max_iters=100)
    s = dp.compile(fns, method="ladmm", device="cpu", linear_solve_config=cfg)      # compiled once, the data is a Placeholder

    def local_solve(loc):
        yy = torch.view_as_complex(loc["y"].contiguous())
        yph.value = yy
        with torch.no_grad():
            out = s.solve(x0=ifft2(yy).real.contiguous(), rhos=0.5, lams=0.03, max_iter=3)
        assert s.last_path == "fused-cg", s.last_path
        return out

    y_ri = torch.view_as_real(torch.from_numpy(y0)).contiguous()
    out = dd.solve_sharded(local_solve, {"y": y_ri} if rank == 0 else None, src=0)
    # per-shard oracle parity
    a, b_ = dd.shard_slices(B, world)[rank]
    shard_err = 0.0
    if b_ > a:
        ys = torch.from_numpy(y0[a:b_])
        mk = torch.from_numpy(mask0)
        op = O.lin_custom(lambda x: mk * O.fft2c(x), lambda yy: O.ifft2c(mk * yy).real)
        terms = [O.sum_squares(op, b=ys), O.nonneg(O.lin_identity()), O.deep_prior(O.lin_identity(), O.FFDNetOracle(wts, per_band=True))]
        with torch.no_grad():
            ref = O.solve(terms, "ladmm", x0=O.ifft2c(ys).real, rhos=0.5, lams=0.03, max_iter=3, linear_solve_config=O.LinearSolveConfig(rtol=1e-6, max_iters=100))
        shard_err = float((out[a:b_] - ref).norm() / ref.norm())
    # shared tables on the Fourier path
    gt2, b2, psf = synthetic.deconv_case(2, 1, 24, 32, seed=9, ksize=7, ksigma=2.0)
    bt = torch.from_numpy(b2)

    def tv(shared):
        xx = dp.Variable()
        sv = dp.compile(dp.sum_squares(dp.conv(xx, psf) - bt) + dp.norm1(dp.grad(xx, dim=0)) + dp.norm1(dp.grad(xx, dim=1)), method="admm", device="cpu")
        if shared:
            tabs = dd.share_tables(sv, (1, 1, 24, 32), src=0, device=torch.device("cpu"))
            assert {"otf0", "t0", "t1", "consts"} <= set(tabs)
            from dprox import _ops as ops
            calls = []
            real = ops.make_otf
            ops.make_otf = lambda *a_, **k_: (calls.append(1), real(*a_, **k_))[1]
            try:
                o = sv.solve(x0=bt, rhos=0.2, lams=0.01, max_iter=3)
            finally:
                ops.make_otf = real
            assert rank == 0 or not calls, "a rank that received the tables rebuilt them"
            return o
        return sv.solve(x0=bt, rhos=0.2, lams=0.01, max_iter=3)
    tab_err = float((tv(True) - tv(False)).abs().max())
    q.put({"rank": rank, "shape": tuple(out.shape), "shard_err": shard_err, "tab_err": tab_err})
    dist.barrier()
    dist.destroy_process_group()


def test_world4_config4_shards_match_their_oracle_and_shared_tables():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    B, world = 6, 4                                  # ragged: shards of 2, 2, 1, 1
    procs = [ctx.Process(target=_worker_c4, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=600) for _ in procs]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    assert sorted(r["rank"] for r in res) == list(range(world))
    for r in res:
        assert r["shape"] == (B, 1, 24, 24)
        assert r["shard_err"] <= 1e-5, r            # every shard is the oracle's answer on that sub-batch
        assert r["tab_err"] == 0.0, r               # broadcast tables = locally built tables


def _worker_abi(rank, world, port, B, q):
    """The C-ABI transport (dprox.distributed.Comm -> dpx_comm_* of csrc/dpx_comm.hip) with world > 1: the library binds the
    shared-memory stand-in for librccl of tests/emul (dpx_comm_use_library), so the direct-send all-gather, the root's sends of the
    scatter and the broadcast run between real processes -- slice offsets, non-zero roots, payloads larger than the transport's
    ring, ragged and empty shards through solve_sharded, tables shared through the communicator."""
    import sys
    for p in (ROOT, os.path.join(ROOT, "delta-prox_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)      # host side only: ships the unique id and the shapes
    import emul_util
    be = emul_util.use_emulator()
    import synthetic
    import dprox as dp
    from dprox import distributed as dd
    L = be.lib()
    L.call("dpx_comm_use_library", os.path.join(emul_util.EMUL, "librccl_stub.so").encode())
    comm = dd.Comm.from_process_group()
    assert (comm.rank, comm.world) == (rank, world)
    assert L.query("dpx_comm_rank", comm._h) == rank and L.query("dpx_comm_world", comm._h) == world
    calls = {}
    real_call = L.call

    def counting_call(name, *a):
        calls[name] = calls.get(name, 0) + 1
        return real_call(name, *a)
    L.call = counting_call
    res = {"rank": rank}
    # ---- the three collectives themselves --------------------------------------------------------------------------------
    n_big = 300_000                                      # 1.2 MB per rank: several turns of the transport's 256 KB rings
    mine = (torch.arange(n_big, dtype=torch.float32) * (rank + 1) + rank).view(2, -1)
    for ring in (0, 1):                                  # world - 1 direct sends (default) and ncclAllGather
        with be.tuned(comm_allgather_ring=ring):
            got = comm.all_gather(mine)
        want = torch.cat([(torch.arange(n_big, dtype=torch.float32) * (r + 1) + r).view(2, -1) for r in range(world)], 0)
        assert torch.equal(got, want), ("all_gather", ring)
    for root in (0, world - 1):
        t = torch.full((1000, 7), float(root + 5)) if rank == root else torch.zeros(1000, 7)
        comm.broadcast(t, root)
        assert bool((t == root + 5).all()), ("broadcast", root)
        full = torch.arange(world * 3 * 50, dtype=torch.float32).view(world * 3, 50) + root if rank == root else None
        part = comm.scatter(full, 3, (50,), torch.float32, torch.device("cpu"), root)
        want = (torch.arange(world * 3 * 50, dtype=torch.float32).view(world * 3, 50) + root)[rank * 3:(rank + 1) * 3]
        assert torch.equal(part, want), ("scatter", root)
    assert L.query("dpx_comm_broadcast", comm._h, None, 16, world, None) < 0            # root out of range: an error, not a hang
    # ---- the sharded solve over this transport (ragged / empty shards) ---------------------------------------------------
    gt, b, psf0 = synthetic.deconv_case(B, 1, 24, 32, seed=3, ksize=7, ksigma=2.0)
    consts = dd.broadcast_constants({"psf": torch.from_numpy(psf0)} if rank == 0 else None, src=0, comm=comm)
    psf = consts["psf"].numpy()

    def local_solve(loc):
        x = dp.Variable()
        fns = dp.sum_squares(dp.conv(x, psf) - loc["b"]) + dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1))
        s = dp.compile(fns, method="admm", device="cpu")
        return s.solve(x0=loc["b"], rhos=0.2, lams=0.01, max_iter=4)

    n0 = dict(calls)
    out = dd.solve_sharded(local_solve, {"b": torch.from_numpy(b)} if rank == 0 else None, src=0, comm=comm)
    res["used_abi"] = (calls.get("dpx_comm_scatter", 0) > n0.get("dpx_comm_scatter", 0) and
                       calls.get("dpx_comm_allgather", 0) > n0.get("dpx_comm_allgather", 0) and
                       calls.get("dpx_comm_broadcast", 0) > 2)
    full = local_solve({"b": torch.from_numpy(b)})
    res["err"], res["shape"] = float((out - full).abs().max()), tuple(out.shape)
    # ---- tables built on rank 0 reach the other ranks through the communicator -------------------------------------------
    bt = torch.from_numpy(b[:1])
    xx = dp.Variable()
    sv = dp.compile(dp.sum_squares(dp.conv(xx, psf) - bt) + dp.norm1(dp.grad(xx, dim=0)) + dp.norm1(dp.grad(xx, dim=1)), method="admm", device="cpu")
    tabs = dd.share_tables(sv, (1, 1, 24, 32), src=0, device=torch.device("cpu"), comm=comm)
    res["tab_err"] = float((sv.solve(x0=bt, rhos=0.2, lams=0.01, max_iter=4) - full[:1]).abs().max())      # (full: tables built locally)
    res["tabs"] = sorted(tabs)
    comm.close()
    q.put(res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 5), (4, 6), (4, 3)])
def test_c_abi_transport_runs_with_more_than_one_rank(world, B):
    import emul_util
    emul_util.build()                                    # (once, before the ranks race to build it)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_abi, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=600) for _ in procs]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    assert sorted(r["rank"] for r in res) == list(range(world))
    for r in res:
        assert r["used_abi"], r                          # scatter, all-gather and broadcast went through dpx_comm_*
        assert r["shape"] == (B, 1, 24, 32)
        assert r["err"] == 0.0, r                        # sharding the direct (Fourier) path is exact
        assert r["tab_err"] == 0.0, r                    # tables received through the communicator = tables built locally
        assert {"otf0", "t0", "t1", "consts"} <= set(r["tabs"])
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("dpx_rccl_stub_")]      # the last rank to leave unlinks the segment


def test_shard_slices():
    from dprox.distributed import shard_slices
    assert shard_slices(8, 8) == [(i, i + 1) for i in range(8)]
    assert shard_slices(8, 4) == [(0, 2), (2, 4), (4, 6), (6, 8)]
    assert shard_slices(5, 2) == [(0, 3), (3, 5)]
    assert shard_slices(1, 2) == [(0, 1), (1, 1)]
    assert shard_slices(32, 8)[-1] == (28, 32)
