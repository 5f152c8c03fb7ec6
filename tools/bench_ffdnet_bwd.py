#!/usr/bin/env python
"""GPU box: FFDNet-colour forward + backward-data (frozen weights, gradients w.r.t. image and sigma) at the config-5 shape, on the
f32-input kernels (compute_mode 'f32': _FFDNetFn) and on the split kernels (_FFDNetSplitFn)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
import synthetic
from dprox.proxfn.pnp.denoisers import FFDNetColorDenoiser
dev = torch.device("cuda")
B, C, H, W = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (4, 3, 512, 512)
den = FFDNetColorDenoiser(synthetic.ffdnet_weights(7)).to(dev)
den.requires_grad_(False)
x0 = torch.rand(B, C, H, W, device=dev)
w = torch.randn_like(x0)
flop = 2 * 9 * (13 * 96 + 10 * 96 * 96 + 96 * 12) * (H // 2) * (W // 2) * B
ref = None
for mode in ("f32", "bf16x3", "f16x2"):
    den.model.compute_mode = mode
    def step():
        x = x0.clone().requires_grad_(True)
        sig = torch.full((B,), 0.05, device=dev, requires_grad=True)
        y = den.denoise(x, sig)
        (y * w).sum().backward()
        return x.grad, sig.grad
    for _ in range(2): g = step()
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 5
    for _ in range(n): g = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    if ref is None: ref = g
    e = float((g[0] - ref[0]).norm() / ref[0].norm()); es = float((g[1] - ref[1]).norm() / ref[1].norm())
    print(f"FFDNet-colour {B}x{C}x{H}x{W} forward + backward-data, mode {mode:7s}: {dt*1e3:7.2f} ms = {2*flop/dt/1e12:6.1f} TFLOP/s (fp32-equivalent, two passes); "
          f"d/dx vs f32 path {e:.2e}, d/dsigma {es:.2e}")
