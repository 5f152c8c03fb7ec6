#!/usr/bin/env python
"""GPU box: where a train_unrolled_pnp step (tools/bench_train.py, frozen denoiser) spends its wall clock: forward / backward / optimiser
phases with a synchronisation between them, against the library's per-kernel event times of the same phases."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "delta-prox_amd"), os.path.join(ROOT, "tools")]
import torch
import dprox as dp, synthetic
from dprox import _backend as be
import bench_train
import torch.nn.functional as F
dev = "cuda"
trainable = len(sys.argv) > 1 and sys.argv[1] == "trainable"
step, m, reg = bench_train.build(dp, synthetic, dev, 2, 768, 10, trainable)
L = be.lib()
def report():
    buf = ctypes.create_string_buffer(1 << 16)
    L.call("dpx_timing_report", buf, len(buf))
    return sum(float(l.split()[2]) for l in buf.value.decode().splitlines() if l.split())
for _ in range(3):
    step()
torch.cuda.synchronize()
# re-implement the step with phase boundaries (same calls as bench_train.build's step)
import types
src = step.__closure__
cells = {n: c.cell_contents for n, c in zip(step.__code__.co_freevars, step.__closure__)}
P, Bv, blur, gt, noise, solver, loop = (cells[k] for k in ("P", "Bv", "blur", "gt", "noise", "solver", "loop"))
for rep in range(3):
    L.call("dpx_timing_enable", 1); report()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    psf = m.get_psf(); P.value = psf
    inp = blur.forward(gt) + noise
    Bv.value = inp
    pred = solver.solve(x0=inp.detach(), rhos=m.rhos, lams={reg: m.lams.sqrt()})
    loss = F.mse_loss(gt, pred)
    torch.cuda.synchronize(); t1 = time.perf_counter(); k_f = report()
    loop.optimizer.zero_grad(set_to_none=True)
    loss.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter(); k_b = report()
    loop.optimizer.step()
    torch.cuda.synchronize(); t3 = time.perf_counter(); k_o = report()
    L.call("dpx_timing_enable", 0)
    print(f"forward {1e3*(t1-t0):7.2f} ms wall / {k_f:7.2f} ms dpx kernels | backward {1e3*(t2-t1):7.2f} / {k_b:7.2f} | optimiser {1e3*(t3-t2):6.2f} / {k_o:5.2f} | "
          f"alloc: {torch.cuda.memory_allocated()/1e9:.2f} GB, reserved {torch.cuda.memory_reserved()/1e9:.2f} GB")
