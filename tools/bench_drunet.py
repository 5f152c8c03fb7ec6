#!/usr/bin/env python
"""DRUNet-colour forward (DRUNetDenoiser incl. its four-quadrant splitting): time and TFLOP/s vs the fp32-MFMA roofline (GPU only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd")]
import torch
from dprox.proxfn.pnp.denoisers import DRUNetDenoiser
import synthetic as O
B, S = (int(sys.argv[1]) if len(sys.argv) > 1 else 8), (int(sys.argv[2]) if len(sys.argv) > 2 else 256)
dev = torch.device("cuda")
den = DRUNetDenoiser(3, O.drunet_weights(21, 4, 3)).to(dev)
x = torch.rand(B, 3, S, S, device=dev)
sig = torch.full((B,), 0.05, device=dev)
with torch.no_grad():
    for _ in range(2): y = den.denoise(x, sig)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 3
    for _ in range(n): y = den.denoise(x, sig)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
nc, nb = (64, 128, 256, 512), 4
def flops(h, w):            # one UNetRes pass on an h x w input
    f = 2 * 9 * (4 * 64 + 64 * 3) * h * w
    for l in range(4):
        hh, ww = h >> l, w >> l
        f += 2 * 9 * nc[l] * nc[l] * hh * ww * 2 * nb * (1 if l == 3 else 2)
    for l in range(3):
        hh, ww = h >> (l + 1), w >> (l + 1)
        f += 2 * 4 * nc[l] * nc[l + 1] * hh * ww * 2          # strided conv down + transposed conv up
    return f
print(f"DRUNet-color B={B} {S}x{S}: {dt*1e3:.2f} ms  {B*flops(S,S)/dt/1e12:.1f} TFLOP/s ({B*flops(S,S)/dt/157.3e12*100:.1f}% of fp32 MFMA peak; single-pass FLOP count)")
