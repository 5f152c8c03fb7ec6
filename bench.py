#!/usr/bin/env python
"""Headline benchmark: ADMM iterations/sec (and PSNR) on the BASELINE.json config-2 workload --
batch-8 3x1024x1024 RGB deconvolution, sum_squares(conv(x,psf)-b) + norm1(grad_H x) + norm1(grad_W x),
FFT-diagonalised x-update + L1-TV prox -- through the drop-in `dprox` API on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE ADMM iteration over one per-GPU batch of 8 images, inputs resident in HBM.  With N
GPUs every rank solves its own independent batch of 8 (weak scaling, no data-path collective: the
images of a batch never interact); `value` = N*K / max-over-ranks time.

The JSON line also carries
  roofline     : dominant kernel's algorithmic bytes per launch / its average duration measured with
                 HIP events on the launch stream (second pass of K iterations with the library's
                 per-kernel timers on), against 8 TB/s HBM3E;
  cpu_baseline : the reference-schedule CPU oracle (oracle/, a port of the reference's PyTorch-CPU
                 path pinned on its golden vectors) timed on this box's host cores on a bounded sample.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "delta-prox_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK = 8.0e12          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
HBM_COPY = 6.29e12
B, C, H, W = 8, 3, 1024, 1024
RHO, LAM = 0.1, 0.005
# algorithmic HBM bytes per element moved by each kernel of the iteration (fp32, N = B*C*H*W elements);
# DESIGN.md section "kernels" derives them.  SURVEY 8(d) whole-iteration figure: 64 B/element.
KERNEL_BYTES_PER_ELEM = {
    "k_rhs": 24.0,            # reads v0,v1,u0,u1,K^T b ; writes rhs
    "k_rows_r2c": 8.0,        # reads rhs ; writes half spectrum
    "k_cols": 8.0,            # reads + writes half spectrum
    "k_rows_c2r": 8.0,        # reads half spectrum ; writes x
    "k_zupdate": 28.0,        # reads x,u0,u1 ; writes v0,v1,u0,u1
    "k_iter_rows_seq": 24.0,  # streaming fused rows (wave per band): same algorithmic traffic as k_iter_rows
    "k_iter_rows": 24.0,      # fused rows: reads spectrum, u0, u1 ; writes u0, u1, spectrum (x, v stay on chip)
    "k_cols_p2": 12.0,        # column solve: reads spectrum + data spectrum ; writes spectrum (denominators are L2-resident)
    "k_rows_r2c_p2": 8.0,
    "k_rows_c2r_p2": 8.0,
}
ITER_BYTES_PER_ELEM = 64.0          # 16 fp32 passes per element for the un-fused 5-kernel schedule
DESIGN_BYTES_PER_ELEM = 36.0        # what the two-kernel iteration actually has to move (k_cols_p2 12 + k_iter_rows 24)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def make_problem(dp, synthetic, rank, device):
    rng = np.random.RandomState(2023 + rank)
    gt = synthetic.synth(rng, B, C, H, W)
    psf = synthetic.point_spread_function(15, 5.0)
    gt_d = torch.from_numpy(gt).to(device)
    blur = dp.conv(dp.Variable(), psf).to(device)          # blur through the backend's own conv
    noise = torch.from_numpy((rng.randn(B, C, H, W) * (2.0 / 255.0)).astype(np.float32)).to(device)
    b = (blur.forward(gt_d) + noise).contiguous()
    x = dp.Variable()
    fns = dp.sum_squares(dp.conv(x, psf) - b) + dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1))
    solver = dp.compile(fns, method="admm", device=device)
    return solver, x, b, gt_d, psf


def psnr_per_image(out, gt):
    mse = ((out - gt) ** 2).reshape(out.shape[0], -1).mean(dim=1)
    return (10.0 * torch.log10(1.0 / mse)).tolist()


def timing_report(be):
    buf = ctypes.create_string_buffer(1 << 16)
    be.lib().call("dpx_timing_report", buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, tot = line.split()
        out[name] = (int(cnt), float(tot))
    return out


def cpu_baseline(b_host, psf, n_iters=4, sample_b=2):
    """reference-schedule oracle on the host cores: `sample_b` of the 8 images, 1 warm-up + n timed iterations"""
    import oracle as O
    bs = b_host[:sample_b].contiguous()
    terms = [O.sum_squares(O.lin_conv(psf).minus(bs)), O.norm1(O.lin_grad(0)), O.norm1(O.lin_grad(1))]
    stamps = []
    O.solve(terms, "admm", x0=bs, rhos=RHO, lams=LAM, max_iter=1 + n_iters,
            callback=lambda **kw: stamps.append(time.perf_counter()))
    per_iter = (stamps[-1] - stamps[0]) / n_iters            # first iteration = warm-up (OTF build, caches)
    its_batch8 = (sample_b / B) / per_iter                   # iterations/s of a batch-8 problem
    return {"value": its_batch8, "unit": "it/s (batch of 8x3x1024x1024)", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{sample_b} of the {B} images, 1 warm-up + {n_iters} timed ADMM iterations of the reference-schedule "
                      f"oracle ({per_iter:.2f} s/iter), scaled x{sample_b}/{B} to the batch-8 rate"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU path)"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=device)

    import dprox as dp
    import synthetic
    from dprox import _backend as be

    solver, xvar, b, gt, psf = make_problem(dp, synthetic, rank, device)
    K, Wm = a.steps, a.warmup

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up (builds twiddles / OTF tables / workspaces, W iterations) ---------------------------
    solver.solve(x0=b, rhos=RHO, lams=LAM, max_iter=max(Wm, 1))
    assert solver.last_path == "fused", "bench must run the fused HIP iteration"

    # ---- timed region: exactly K iterations -------------------------------------------------------------
    x0, rhos, lams, _ = solver.defaults(b, RHO, LAM, K)
    rhos = rhos.to(device)
    lams = {k: v.to(device) for k, v in lams.items()}
    state = solver.initialize(b)
    barrier()
    t0 = time.perf_counter()
    state = solver.iters(state, rhos, lams, K)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- second pass with per-kernel HIP-event timers (roofline leg) -------------------------------------
    be.lib().call("dpx_timing_enable", 1)
    state2 = solver.initialize(b)
    timing_report(be)                                   # drop the initialize() launches
    solver.iters(state2, rhos, lams, K)
    torch.cuda.synchronize()
    rep = timing_report(be)
    be.lib().call("dpx_timing_enable", 0)

    # ---- quality: a clean 50-iteration solve ---------------------------------------------------------------
    out = solver.solve(x0=b, rhos=RHO, lams=LAM, max_iter=50)
    psnr_in, psnr_out = psnr_per_image(b, gt), psnr_per_image(out, gt)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    n_elem = B * C * H * W
    total_ms = sum(t for _, t in rep.values())
    kernels = {k: {"launches": c, "avg_us": 1e3 * t / c, "share": t / total_ms,
                   "GBps": KERNEL_BYTES_PER_ELEM.get(k, 0.0) * n_elem / (1e-3 * t / c) / 1e9}
               for k, (c, t) in rep.items()}
    dom = max(rep, key=lambda k: rep[k][1])
    # HBM traffic of the dominant kernel from the PMC counters: rocprofv3 cannot be run from inside this process, so the
    # figure comes from the committed separate --pmc passes of this same command (tools/pmc_summary.py, profiles/)
    traffic, traffic_src = None, None
    pmc_file = os.path.join(ROOT, "profiles", "r1_pmc_hbm.json")
    if os.path.exists(pmc_file):
        for name, e in json.load(open(pmc_file)).items():
            if name.split("<")[0] == dom and "hbm_traffic_bytes" in e:
                traffic, traffic_src = e["hbm_traffic_bytes"], "profiles/r1_pmc_hbm.json (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, per launch)"
    dom_bytes = KERNEL_BYTES_PER_ELEM.get(dom, 0.0) * n_elem
    dom_avg_s = 1e-3 * rep[dom][1] / rep[dom][0]
    achieved = dom_bytes / dom_avg_s
    it_per_s = world * K / dt
    res = {
        "metric": "admm_iters_per_sec", "value": it_per_s, "unit": "it/s (one iteration = one 8x3x1024x1024 batch)",
        "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": 1e3 * dt / K, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "config 2: batch-8 3x1024x1024 RGB deconv, sum_squares(conv(x,psf)-b)+norm1(grad_H)+norm1(grad_W), "
                               "ADMM rho=0.1 lam=0.005, Gaussian 15/5 PSF",
                   "batch_per_gpu": B, "global_batch": B * world, "shape": [C, H, W], "parallelism": f"batch-shard x{world}"},
        "psnr_db": {"input_mean": float(np.mean(psnr_in)), "admm50_mean": float(np.mean(psnr_out)), "admm50_per_image": psnr_out},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK, "frac_of_measured_copy": achieved / HBM_COPY, "traffic": traffic,
                     "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_us": dom_avg_s * 1e6},
        "roofline_iteration": {"algorithmic_bytes_per_iter": ITER_BYTES_PER_ELEM * n_elem,
                               "achieved_GBps": (it_per_s / world) * ITER_BYTES_PER_ELEM * n_elem / 1e9,
                               "frac": (it_per_s / world) * ITER_BYTES_PER_ELEM * n_elem / HBM_PEAK,
                               "note": "SURVEY 8(d) accounting (64 B/element/iteration); the fused two-kernel schedule moves "
                                       "36 B/element, see design_bytes_*",
                               "design_bytes_per_iter": DESIGN_BYTES_PER_ELEM * n_elem,
                               "design_achieved_GBps": (it_per_s / world) * DESIGN_BYTES_PER_ELEM * n_elem / 1e9,
                               "design_frac": (it_per_s / world) * DESIGN_BYTES_PER_ELEM * n_elem / HBM_PEAK},
        "kernels": kernels,
    }
    if world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(b.cpu(), psf)
    print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
