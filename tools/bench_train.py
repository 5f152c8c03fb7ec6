#!/usr/bin/env python
"""The reference's one published throughput figure (notebooks/quickstart.ipynb:254-257: 1.49 it/s, GPU unstated): one training
step of the end-to-end optics pipeline -- a batch of 2 RGB patches of 768 x 768 (contrib/optic/utils.py:158-166), ADMM unrolled
10 times (specialize 'unroll', algo/specialization/unroll.py:21-58) on  sum_squares(conv_doe(x, PSF), b) + deep_prior(x, 'ffdnet_color'),
MSE loss, backward through the solver, AdamW step (algo/primitives.py:125-205).  Trainable as in the notebook: the PSF that the
optics model hands over (here a learnable 3 x 768 x 768 PSF tensor stands in for the wave-propagation model, which is outside this
backend's scope), the rho_t and sigma_t schedules; the denoiser's weights are frozen (deep_prior's default trainable=False) --
`trainable=True` (weight gradients of all 12 layers every unrolled iteration) is timed next to it.

    python tools/bench_train.py [--bs 2] [--size 768] [--iters 10] [--steps 5] [--kernels]

Also the `train_unrolled_pnp` leg of bench.py (imported from there).  Seeded FFDNet weights (no checkpoint download without a network).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "delta-prox_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def build(dp, synthetic, device, bs, size, iters, trainable, seed=2024, nc=96, nb=12):
    """-> (step, modules): step() runs one optimisation step and returns the loss tensor"""
    from dprox.algo.training import TrainLoop
    from dprox.proxfn.pnp.denoisers import FFDNet, FFDNetColorDenoiser
    rng = np.random.RandomState(seed)
    gt = torch.from_numpy(synthetic.synth_detail(rng, bs, 3, size, size) if size >= 256 else synthetic.synth(rng, bs, 3, size, size)).to(device)
    k = 15 if size >= 64 else 5
    psf0 = synthetic.point_spread_function(k, 5.0 if size >= 64 else 1.2)            # [k, k, 1]
    full = np.zeros((1, 3, size, size), np.float32)                                  # the optics model hands over a sensor-sized PSF per colour
    full[:, :, :k, :k] = psf0[:, :, 0]
    full = np.roll(full, (size // 2 - k // 2, size // 2 - k // 2), axis=(-2, -1))     # centred, as psf2otf2 expects it (linop/conv.py:59-95)

    class Optics(torch.nn.Module):
        """the trained part: PSF (stand-in for the DOE model's output), rho_t, sigma_t"""

        def __init__(self):
            super().__init__()
            rhos, sig = dp.log_descent(49, 7.65, iters, sigma=7.65 / 255)
            self.psf = torch.nn.Parameter(torch.from_numpy(full))
            self.rhos = torch.nn.Parameter(rhos.float())
            self.lams = torch.nn.Parameter(sig.float())             # sigma_t^2; the solver gets its square root (quickstart notebook)

        def get_psf(self):
            p = self.psf.clamp_min(0)
            return p / p.sum(dim=(-2, -1), keepdim=True)

    m = Optics().to(device)
    den = FFDNetColorDenoiser(synthetic.ffdnet_weights(7)) if (nc, nb) == (96, 12) else None
    if den is None:                                                                   # (tiny network for the emulator's smoke run)
        den = FFDNetColorDenoiser()
        den.model = FFDNet(in_nc=3, out_nc=3, nc=nc, nb=nb, act_mode="R")
        den.model.load_layers(synthetic.ffdnet_weights(7, 3, 3, nc, nb))
    x, P, Bv = dp.Variable(), dp.Placeholder(), dp.Placeholder()
    reg = dp.deep_prior(x, denoiser=den, trainable=trainable)
    solver = dp.compile(dp.sum_squares(dp.conv_doe(x, P, circular=True), Bv) + reg, method="admm", device=device)
    solver = dp.specialize(solver, method="unroll", device=device, max_iter=iters)
    blur = dp.conv_doe(dp.Variable(), P, circular=True).to(device)
    params = torch.nn.ModuleList([m] + ([reg.denoiser] if trainable else []))
    loop = TrainLoop(params, lr=1e-4, weight_decay=1e-3, savedir=os.path.join("/tmp", f"dpx_bench_train_{os.getpid()}"))
    noise = torch.from_numpy((rng.randn(bs, 3, size, size) * 7.65 / 255).astype(np.float32)).to(device)

    def step():
        psf = m.get_psf()
        P.value = psf
        inp = blur.forward(gt) + noise                      # differentiable w.r.t. the PSF like the notebook's img_psf_conv
        Bv.value = inp
        pred = solver.solve(x0=inp.detach(), rhos=m.rhos, lams={reg: m.lams.sqrt()})
        return loop.step(gt, pred)

    return step, m, reg


def run(dp, synthetic, be, device, bs=2, size=768, iters=10, steps=5, trainable=False, kernels=False, **kw):
    step, m, reg = build(dp, synthetic, device, bs, size, iters, trainable, **kw)
    sync = torch.cuda.synchronize if torch.cuda.is_available() else (lambda: None)
    l0, _ = step()
    step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        ll, _ = step()
    sync()
    dt = (time.perf_counter() - t0) / steps
    out = {"workload": f"{bs}x3x{size}x{size}, ADMM unrolled x{iters}: sum_squares(conv_doe(x, PSF), b) + deep_prior(ffdnet_color, "
                       f"{'trainable weights' if trainable else 'frozen weights (the notebook default)'}), MSE, backward, AdamW step "
                       "(PSF, rho_t, sigma_t" + (", 852k denoiser weights" if trainable else "") + ")",
           "ms_per_step": dt * 1e3, "steps_per_s": 1 / dt, "loss_first": l0, "loss_last": ll,
           "denoiser_arithmetic": getattr(reg.denoiser.model, "compute_mode", "?") if not trainable else "f32 (MFMA f32-input kernels: differentiable w.r.t. the weights)",
           "psf_grad_norm": float(m.psf.grad.norm()) if m.psf.grad is not None else None,
           "rho_grad_norm": float(m.rhos.grad.norm()) if m.rhos.grad is not None else None}
    if kernels:
        import ctypes
        be.lib().call("dpx_timing_enable", 1)
        buf = ctypes.create_string_buffer(1 << 16)
        be.lib().call("dpx_timing_report", buf, len(buf))
        step()
        sync()
        be.lib().call("dpx_timing_report", buf, len(buf))
        be.lib().call("dpx_timing_enable", 0)
        rows = []
        for line in buf.value.decode().splitlines():
            name, cnt, tot = line.split()
            rows.append((float(tot), int(cnt), name))
        rows.sort(reverse=True)
        tot = sum(r[0] for r in rows)
        out["kernels_ms_per_step"] = {n: {"launches": c, "ms": round(t, 3)} for t, c, n in rows[:14]}
        out["kernel_ms_total"] = tot
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=2)
    ap.add_argument("--size", type=int, default=768)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--kernels", action="store_true")
    ap.add_argument("--emul", action="store_true", help="smoke run on the SIMT emulator (tiny sizes)")
    a = ap.parse_args()
    import json
    if a.emul:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import emul_util
        emul_util.use_emulator()
    import dprox as dp
    import synthetic
    from dprox import _backend as be
    dev = "cpu" if a.emul else "cuda"
    kw = dict(nc=16, nb=3) if a.emul else {}
    for tr in (False, True):
        print(json.dumps(run(dp, synthetic, be, dev, a.bs, a.size, a.iters, a.steps, trainable=tr, kernels=a.kernels, **kw), indent=1))
