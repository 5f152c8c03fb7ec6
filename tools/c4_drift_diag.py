import sys
sys.path[:0] = ['/root/repo', '/root/repo/delta-prox_amd', '/root/repo/tests']
import numpy as np, torch
import dprox as dp, synthetic
from dprox import _backend as be
from dprox.contrib import masked_fft
from dprox.linalg import LinearSolveConfig
from dprox.utils import ifft2
import oracle as O
from dprox.proxfn.pnp.denoisers import FFDNetDenoiser
g = np.load('/root/repo/tests/golden/g32c_full_c4_trajectory.npz')
dev = 'cuda'
gt, mask, y = synthetic.csmri_case(4, 320, 320, seed=int(g["seed"]), center=32)
mask, y = torch.from_numpy(mask).to(dev), torch.from_numpy(y).to(dev)
def rel(a, b): return float(np.linalg.norm((a.astype(np.float64) - b).ravel()) / np.linalg.norm(b.ravel().astype(np.float64)))
def run(tag, mode=None, knobs=None):
    den = FFDNetDenoiser(O.ffdnet_weights(11, 1, 1, 64, 15)).to(dev)
    if mode: den.model.compute_mode = mode
    x = dp.Variable()
    fns = dp.sum_squares(masked_fft(x, mask), y) + dp.nonneg(x) + dp.deep_prior(x, denoiser=den)
    x0 = ifft2(y).real.float().contiguous()
    import contextlib
    ctx = be.tuned(**knobs) if knobs else contextlib.nullcontext()
    with torch.no_grad(), ctx:
        s = dp.compile(fns, method="ladmm", device=dev, linear_solve_config=LinearSolveConfig(rtol=1e-6, max_iters=100))
        out = {}
        for it in (5, 10):
            xo = s.solve(x0=x0, rhos=0.5, lams=0.03, max_iter=it)
            out[it] = xo[..., ::4, ::4].cpu().numpy()
        its = [int(v) for v in s.least_square.cg_iters[-10:]]
    print(f"{tag:40s} it5: vs ref {rel(out[5], g['it5_x']):.2e} vs f64 {rel(out[5], g['it5_x_f64']):.2e} | it10: vs ref {rel(out[10], g['x']):.2e} vs f64 {rel(out[10], g['x_f64']):.2e}  cg {its}", flush=True)
print("reference: it5 vs f64", rel(g['it5_x'], g['it5_x_f64']), " it10 vs f64", rel(g['x'], g['x_f64']))
run("default (f16x2)")
run("bf16x3", mode="bf16x3")
run("f32 denoiser", mode="f32")
run("no fold", knobs=dict(pnp_cg_no_fold=1))
run("no hint", knobs=dict(cg_no_hint=1))
run("cg unfused", knobs=dict(cg_unfused=1))
