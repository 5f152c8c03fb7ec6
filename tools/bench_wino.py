#!/usr/bin/env python
"""FFDNet forward, direct split-f16 layers ("f16x2") against the Winograd ones ("f16x2w"): accuracy against the f32-input mode and time per
forward with per-kernel device times (GPU only).  python tools/bench_wino.py [B]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
from dprox import _backend as be
from dprox.proxfn.pnp.denoisers import FFDNetColorDenoiser, FFDNetDenoiser
import synthetic as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
L = be.lib()
buf = ctypes.create_string_buffer(1 << 16)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@torch.no_grad()
def run(tag, den, x, sig, flop):
    outs = {}
    for mode in ("f32", "f16x2", "f16x2w"):
        den.model.compute_mode = mode
        den.model.f16_fallback = "raise"
        try:
            for _ in range(2):
                y = den.denoise(x, sig)
        except be.F16RangeError:
            pass                                          # (probe builds produce garbage on purpose)
        torch.cuda.synchronize()
        L.call("dpx_timing_enable", 1)
        L.call("dpx_timing_report", buf, len(buf))
        n = 5
        t0 = time.perf_counter()
        be._f16_pending = False
        for _ in range(n):
            y = den.model(x, sig) if x.shape[1] == den.model.in_nc else den.denoise(x, sig)
        torch.cuda.synchronize()
        be._f16_pending = False
        L.query("dpx_ffdnet_f16_overflow", 1)
        dt = (time.perf_counter() - t0) / n
        L.call("dpx_timing_report", buf, len(buf))
        L.call("dpx_timing_enable", 0)
        outs[mode] = y.clone()
        print(f"{tag} mode={mode:7s}: {dt * 1e3:8.3f} ms   {flop / dt / 1e12:7.1f} TFLOP/s fp32-equivalent   rel-L2 vs f32 mode {rel(y, outs['f32']):.2e}   "
              f"max-abs/max {float((y - outs['f32']).abs().max() / outs['f32'].abs().max()):.2e}")
        if mode != "f32":
            print("   " + buf.value.decode().replace("\n", "\n   "))


den = FFDNetColorDenoiser(O.ffdnet_weights(7)).to(dev)
x = torch.rand(B, 3, 1024, 1024, device=dev)
run(f"FFDNet-color B={B} 1024^2", den, x, torch.full((B,), 0.05, device=dev), 2 * 9 * (13 * 96 + 10 * 96 * 96 + 96 * 12) * 512 * 512 * B)
for Bg in (4, 32):
    g = FFDNetDenoiser(O.ffdnet_weights(11, 1, 1, 64, 15)).to(dev)
    xg = torch.rand(Bg, 1, 320, 320, device=dev)
    run(f"FFDNet-gray B={Bg} 320^2", g, xg, torch.full((Bg,), 0.05, device=dev), 2 * 9 * (5 * 64 + 13 * 64 * 64 + 64 * 4) * 160 * 160 * Bg)
