#!/usr/bin/env python
"""FFDNet-color forward at the config-3 shape in the three arithmetic modes: time, effective fp32 TFLOP/s, error vs the f32 path."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
from dprox import _backend as be
from dprox.proxfn.pnp.denoisers import FFDNetColorDenoiser, FFDNetDenoiser
import synthetic as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
for name, den, C, hw in (("color", FFDNetColorDenoiser(O.ffdnet_weights(7)).to(dev), 3, 1024), ("gray", FFDNetDenoiser(O.ffdnet_weights(11, 1, 1, 64, 15)).to(dev), 1, 320)):
    Bn = B if name == "color" else 32
    x = torch.rand(Bn, C, hw, hw, device=dev)
    sig = torch.full((Bn,), 0.05, device=dev)
    ref = None
    for mode in ("f32", "bf16x3", "f16x2", "bf16"):
        den.model.compute_mode = mode
        with torch.no_grad():
            for _ in range(2): y = den.denoise(x, sig)
            torch.cuda.synchronize()
            t0 = time.perf_counter(); n = 3
            for _ in range(n): y = den.denoise(x, sig)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        if ref is None: ref = y
        err = float((y - ref).norm() / ref.norm())
        nc, nb, inc = (96, 12, 3) if name == "color" else (64, 15, 1)
        flop = 2 * 9 * ((4 * inc + 1) * nc + (nb - 2) * nc * nc + nc * 4 * inc) * (hw // 2) ** 2 * Bn
        print(f"FFDNet-{name} B={Bn} {hw}^2 mode={mode:7s}: {dt*1e3:8.2f} ms  {flop/dt/1e12:7.1f} TFLOP/s (fp32-equivalent)  rel-L2 vs f32 path {err:.2e}")
