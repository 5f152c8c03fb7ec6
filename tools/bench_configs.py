#!/usr/bin/env python
"""Timing of the other BASELINE.json configurations (1, 3, 4) through the drop-in API on one MI355X (GPU only).
bench.py is the contract benchmark (config 2); this script fills the table in DESIGN.md."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import numpy as np, torch
import dprox as dp
from dprox.linalg import LinearSolveConfig
from dprox.proxfn.pnp.denoisers import FFDNetColorDenoiser, FFDNetDenoiser
from dprox.utils import fft2, ifft2
import synthetic as O       # seeded weight generators
import synthetic

dev = torch.device("cuda")
which = sys.argv[1:] or ["c1", "c3", "c4"]

def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out

def psnr(a, b): return float(10 * torch.log10(1.0 / ((a - b) ** 2).mean()))

if "c1" in which:
    gt, b, psf = synthetic.deconv_case(1, 1, 256, 256, seed=2023)
    bt = torch.from_numpy(b).to(dev); x = dp.Variable()
    s = dp.compile(dp.sum_squares(dp.conv(x, psf) - bt) + dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1)), method="admm", device=dev)
    dt, out = timed(lambda: s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=20), 20)
    print(f"config1 1x1x256x256 TV-deconv ADMM 20 it: {dt*1e3:.3f} ms/solve = {20/dt:.0f} it/s, PSNR {psnr(torch.from_numpy(b), torch.from_numpy(gt)):.2f} -> {psnr(out.cpu(), torch.from_numpy(gt)):.2f} dB")

if "c3" in which:
    B = 8
    rng = np.random.RandomState(2023); gt = synthetic.synth(rng, B, 3, 1024, 1024); psf = synthetic.point_spread_function(15, 5.0)
    gt_d = torch.from_numpy(gt).to(dev)
    b = (dp.conv(dp.Variable(), psf).to(dev).forward(gt_d) + torch.from_numpy((rng.randn(B, 3, 1024, 1024) * 2 / 255).astype(np.float32)).to(dev)).contiguous()
    x = dp.Variable(); prior = dp.deep_prior(x, denoiser=FFDNetColorDenoiser(O.ffdnet_weights(7)))
    s = dp.compile(dp.sum_squares(dp.conv(x, psf) - b) + prior, method="admm", device=dev)
    rhos, sig = dp.log_descent(35, 5, 30)
    with torch.no_grad():
        dt, out = timed(lambda: s.solve(x0=b, rhos=rhos, lams={prior: sig}, max_iter=30), 1)
    flop = 3.5695e12
    print(f"config3 8x3x1024x1024 PnP(FFDNet-color, random weights) ADMM 30 it: {dt/30*1e3:.2f} ms/it = {30/dt:.1f} it/s; "
          f"denoiser FLOP rate {flop*30/dt/1e12:.1f} TFLOP/s = {flop*30/dt/157.3e12*100:.0f}% of fp32 MFMA peak; path={s.last_path}")

if "c4" in which or "c4s" in which:
    B, H, W = (4 if "c4s" in which else 32), 320, 320        # c4s: one GPU's shard of the batch
    gt, mask, y = synthetic.csmri_case(B, H, W, seed=2023)
    mask_d, y_d = torch.from_numpy(mask).to(dev), torch.from_numpy(y).to(dev)
    class MaskedFFT(dp.LinOp):
        def __init__(self, arg, mask): super().__init__([arg]); self.mask = mask
        def forward(self, x, **kw): return (self.mask * fft2(x)).contiguous()
        def adjoint(self, v, **kw): return ifft2(self.mask * v).real.contiguous()
    from dprox.contrib import masked_fft
    x = dp.Variable()
    fns = dp.sum_squares(masked_fft(x, mask_d), y_d) + dp.nonneg(x) + dp.deep_prior(x, denoiser=FFDNetDenoiser(O.ffdnet_weights(11, 1, 1, 64, 15)))
    s = dp.compile(fns, method="ladmm", device=dev, linear_solve_config=LinearSolveConfig(rtol=1e-6, max_iters=100))
    x0 = ifft2(y_d).real.contiguous()
    with torch.no_grad():
        dt, out = timed(lambda: s.solve(x0=x0, rhos=0.5, lams=0.03, max_iter=10), 1)
    print(f"config4 {B}x1x320x320 CS-MRI LADMM+CG(<=100) + nonneg + FFDNet-gray, 10 outer it: {dt/10*1e3:.3f} ms/outer it, CG its {s.least_square.cg_iters[-10:]}, "
          f"(the seeded random-weight 'denoiser' is not a denoiser: output quality is meaningless here; parity is pinned by fixture G7)")

if "c5" in which:
    # config 5: unrolled ADMM (10 iterations) training step on 4x3x512x512 -- forward + backward w.r.t. the rho / lambda schedules
    gt, b, psf = synthetic.deconv_case(4, 3, 512, 512, seed=2023)
    bt, gtt = torch.from_numpy(b).to(dev), torch.from_numpy(gt).to(dev)
    x = dp.Variable()
    n0, n1 = dp.norm1(dp.grad(x, dim=0)), dp.norm1(dp.grad(x, dim=1))
    s = dp.compile(dp.sum_squares(dp.conv(x, psf) - bt) + n0 + n1, method="admm", device=dev)
    s = dp.specialize(s, method="unroll", device=dev, max_iter=10)
    pdev = "cpu" if os.environ.get("DPX_C5_CPU_PARAMS") else dev      # the trained schedules live on the GPU (nn.Parameters of the unrolled solver)
    rhos = torch.full((10,), 0.1, requires_grad=True, device=pdev)
    l0, l1 = torch.full((10,), 0.005, requires_grad=True, device=pdev), torch.full((10,), 0.005, requires_grad=True, device=pdev)
    def step():
        for p in (rhos, l0, l1):
            p.grad = None
        out = s.solve(x0=bt, rhos=rhos, lams={n0: l0, n1: l1})
        loss = ((out - gtt) ** 2).mean()
        loss.backward()
        return loss
    dt, loss = timed(step, 5)
    if os.environ.get("DPX_C5_PROFILE"):
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable(); step(); torch.cuda.synchronize(); pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(30)
    with torch.no_grad():
        dtf, _ = timed(lambda: s.solve(x0=bt, rhos=rhos.detach(), lams={n0: l0.detach(), n1: l1.detach()}), 5)
    print(f"config5 4x3x512x512 unrolled ADMM x10, MSE loss: fwd+bwd {dt*1e3:.2f} ms/step ({1/dt:.1f} steps/s), inference-only forward {dtf*1e3:.2f} ms; "
          f"loss {float(loss.detach()):.5f}, |g_rho| {float(rhos.grad.abs().sum()):.3e}, |g_lam| {float(l0.grad.abs().sum() + l1.grad.abs().sum()):.3e}")
