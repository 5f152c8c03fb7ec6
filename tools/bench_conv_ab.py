#!/usr/bin/env python
"""A/B of a tuning knob on the FFDNet forward passes bench.py's configs 3 and 4 run (GPU only): alternating runs in one process.
python tools/bench_conv_ab.py conv_double_tile 0 1"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
from dprox import _backend as be
from dprox.proxfn.pnp.denoisers import FFDNetColorDenoiser, FFDNetDenoiser
import synthetic as O
knob, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
dev = torch.device("cuda")
cases = [("colour 8x3x1024^2", FFDNetColorDenoiser(O.ffdnet_weights(7)).to(dev), torch.rand(8, 3, 1024, 1024, device=dev)),
         ("gray 4x1x320^2", FFDNetDenoiser(O.ffdnet_weights(11, 1, 1, 64, 15)).to(dev), torch.rand(4, 1, 320, 320, device=dev)),
         ("gray 32x1x320^2", FFDNetDenoiser(O.ffdnet_weights(11, 1, 1, 64, 15)).to(dev), torch.rand(32, 1, 320, 320, device=dev))]
with torch.no_grad():
    for tag, den, x in cases:
        sig = torch.full((x.shape[0],), 0.05, device=dev)
        outs = {}
        for rnd in range(3):
            for v in vals:
                with be.tuned(**{knob: v}):
                    for _ in range(2):
                        y = den.denoise(x, sig)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    n = 10
                    for _ in range(n):
                        y = den.denoise(x, sig)
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / n
                outs.setdefault(v, y.clone())
                print(f"{tag:20s} {knob}={v}: {dt * 1e3:8.3f} ms", flush=True)
        ref = outs[vals[0]]
        print(f"{tag:20s} bit-identical across settings: {all(torch.equal(ref, o) for o in outs.values())}")
