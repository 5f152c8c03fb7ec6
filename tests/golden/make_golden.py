#!/usr/bin/env python
"""Golden-vector generator -- runs ONLY in the build container (needs /root/reference).

Imports the real Delta-Prox reference (PyTorch-CPU path) through a stub for its
non-arithmetic third-party imports and stores inputs + reference outputs as small
``.npz`` fixtures next to this file.  Nothing here is
reference source: the fixtures are data, the shim only mocks I/O / plotting / RL modules
that are absent from the image, and every number is produced by the reference's own code
+ torch/numpy.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

The pretrained FFDNet checkpoints cannot be downloaded (no network), so the denoiser
cases use seeded random weights (``oracle.ffdnet_weights``) loaded into the reference's
own ``FFDNet`` module.
"""
import contextlib
import importlib.abc
import importlib.machinery
import os
import sys
from unittest.mock import MagicMock

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("DPROX_REFERENCE", "/root/reference")

# ---- import shim: mock only non-arithmetic third-party modules ----
_MISSING = ["imageio", "skimage", "cv2", "munch", "termcolor", "tensorboardX", "cvxpy", "proximal",
            "torchlight", "torchlights", "tfpnp", "graphviz", "IPython"]


class _Loader(importlib.abc.Loader):
    def create_module(self, spec):
        m = MagicMock(name=spec.name)
        m.__name__, m.__path__, m.__spec__, m.__loader__ = spec.name, [], spec, self
        return m

    def exec_module(self, module):
        pass


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _MISSING:
            return importlib.machinery.ModuleSpec(fullname, _Loader(), is_package=True)


sys.meta_path.insert(0, _Finder())
import numpy as np  # noqa: E402
import scipy  # noqa: E402
import scipy.misc  # noqa: E402

_rng0 = np.random.RandomState(0)
scipy.misc.face = lambda gray=False: (_rng0.rand(768, 1024, 3) * 255).astype("uint8")
scipy.misc.ascent = lambda: (_rng0.rand(512, 512) * 255).astype("uint8")
if not hasattr(scipy, "finfo"):
    scipy.finfo = np.finfo
for _n, _t in (("int", int), ("bool", bool), ("float", float)):
    if not hasattr(np, _n):
        setattr(np, _n, _t)

sys.path.insert(0, REF)        # the reference's `dprox`
sys.path.insert(1, ROOT)       # synthetic.py, oracle (weights generator only)
import torch  # noqa: E402
import dprox as dp  # noqa: E402   (the REFERENCE)
from dprox.linalg import LinearSolveConfig  # noqa: E402
from dprox.linalg.solve import cg as ref_cg  # noqa: E402
from dprox.proxfn.pnp.denoisers.base import Denoiser, Denoiser2D  # noqa: E402
from dprox.proxfn.pnp.denoisers.models.network_ffdnet import FFDNet  # noqa: E402
from dprox.utils import fft2, ifft2  # noqa: E402
from dprox.algo.tune.dpir import log_descent  # noqa: E402

import synthetic  # noqa: E402
from oracle.dprox_oracle import admm_f64, ffdnet_weights  # noqa: E402

assert dp.__file__.startswith(REF), dp.__file__
torch.manual_seed(0)
torch.set_num_threads(8)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name:28s} {os.path.getsize(path) / 1024:8.1f} KiB  keys={list(out)}")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def gauss(k, s):
    return synthetic.point_spread_function(k, s)


# --------------------------------------------------------------------------------------
# the REFERENCE evaluated in float64 (the yardstick of every comparison the tests accept above 1e-5)
# --------------------------------------------------------------------------------------
@contextlib.contextmanager
def reference_in_float64():
    """The reference has no float64 mode: conv.forward / adjoint (linop/conv.py:34,40) and least_squares.solve_direct
    (proxfn/sum_square.py:156) end with `.float()`, and to_ndarray casts NumPy kernels to float32 (utils/misc.py:143).  Inside this
    block `torch.Tensor.float` is a no-op on float64 tensors -- the ONLY patch, generator-side, undone on exit -- and the callers
    below hand the reference float64 tensors (observation, x0, the point-spread function as a TENSOR, which to_ndarray passes
    through unchanged, so psf2otf runs in complex128).  Everything else is the reference's own code: the iterate it would produce
    if its casts were not there.  Scalar rho / lambda still become float32 tensors inside Algorithm.defaults (algo/base.py:212-217),
    i.e. the float32-rounded values, promoted exactly."""
    real = torch.Tensor.float
    torch.Tensor.float = lambda self, *a, **k: self if self.dtype == torch.float64 else real(self, *a, **k)
    try:
        yield
    finally:
        torch.Tensor.float = real


def T64(a):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64)))


def ref_f64_admm(b, psf, K, rhos=0.1, lams=0.005, dims=(0, 1), prior=None, extra=None, full=False, callback=None):
    """ADMM on  sum_squares(conv(x, psf) - b) + [TV terms along `dims`] + [deep_prior(denoiser=prior)] + [extra(x)]  run by the REFERENCE in
    float64 (reference_in_float64): returns x, or the full state with full=True."""
    with reference_in_float64():
        x = dp.Variable()
        b64 = T64(b)
        fns = dp.sum_squares(dp.conv(x, T64(psf)) - b64)
        for d in dims:
            fns = fns + dp.norm1(dp.grad(x, dim=d))
        lam_arg = lams
        if prior is not None:
            pf = dp.deep_prior(x, denoiser=prior.double())
            fns = fns + pf
            lam_arg = {pf: lams}
        if extra is not None:
            ef = extra(x)
            fns = fns + ef
            lam_arg = dict(lam_arg)
            lam_arg[ef] = 0.0
        with torch.no_grad():
            out = dp.Problem(fns).solve(method="admm", device="cpu", x0=b64.clone(), rhos=rhos, lams=lam_arg, max_iter=K, return_full_states=full,
                                        callback=callback)
    assert (out[0] if full else out).dtype == torch.float64
    return out


def f64_pin(out, key, ref64, ours64, tol=1e-12):
    """records how far the builder's float64 restatement (oracle.admm_f64) is from the reference's float64 iterate: <= 1e-12 or the
    generator stops (tests/test_oracle_golden.py re-checks the small cases and reads this figure for the large ones).  The
    plug-and-play cases get 1e-10: their x-update divides by |H|^2 + rho with rho down to 1e-5 (log_descent), which amplifies the
    last-bit differences of two float64 FFT call sequences by 1 / min(denominator) -- in float64 as in float32."""
    rel = float((torch.as_tensor(ours64) - ref64).norm() / ref64.norm())
    print(f"   float64: oracle.admm_f64 vs the reference in float64 [{key}]: rel-L2 {rel:.2e}")
    assert rel <= tol, (key, rel, tol)
    out[key + "_oracle_rel"] = np.float64(rel)


# --------------------------------------------------------------------------------------
# reference denoiser wrappers holding seeded weights
# --------------------------------------------------------------------------------------
def load_ffdnet(in_nc, out_nc, nc, nb, seed, gain=0.5):
    net = FFDNet(in_nc=in_nc, out_nc=out_nc, nc=nc, nb=nb, act_mode="R")
    layers = ffdnet_weights(seed, in_nc, out_nc, nc, nb, gain=gain)
    sd = {}
    for i, (w, b) in enumerate(layers):
        sd[f"model.{2 * i}.weight"] = T(w)
        sd[f"model.{2 * i}.bias"] = T(b)
    net.load_state_dict(sd, strict=True)
    return net


class ColorDen(Denoiser):
    def __init__(self, seed=7):
        super().__init__()
        self.model = load_ffdnet(3, 3, 96, 12, seed)

    def _denoise(self, x, sigma):
        return self.model(x, sigma)


class GrayDen(Denoiser2D):
    def __init__(self, seed=11):
        super().__init__()
        self.model = load_ffdnet(1, 1, 64, 15, seed)

    def _denoise(self, x, sigma):
        return self.model(x, sigma)


class MaskedFFT(dp.LinOp):
    """Config-4 operator expressed through the reference's plugin surface (SURVEY 8(a) NB)."""

    def __init__(self, arg, mask):
        super().__init__([arg])
        self.mask = mask

    def forward(self, x, **kw):
        return self.mask * fft2(x)

    def adjoint(self, y, **kw):
        return ifft2(self.mask * y).real


# --------------------------------------------------------------------------------------
def g1_linops():
    rng = np.random.RandomState(101)
    for tag, shape in (("a", (2, 3, 32, 48)), ("b", (1, 1, 15, 21)), ("c", (2, 3, 20, 24))):
        x = T(rng.rand(*shape).astype("float32"))
        y = T(rng.randn(*shape).astype("float32"))
        psf = gauss(7, 2.0)
        v = dp.Variable()
        out = {"x": x, "y": y, "psf": psf}
        c = dp.conv(v, psf)
        out["conv_fwd"], out["conv_adj"], out["conv_diag"] = c.forward(x), c.adjoint(y), c.get_diag(x, freq=True)
        dims = (0, 1, 2) if shape[1] == 3 else (0, 1)
        for d in dims:
            g = dp.grad(v, dim=d)
            out[f"grad{d}_fwd"], out[f"grad{d}_adj"] = g.forward(x), g.adjoint(y)
            out[f"grad{d}_diag"] = g.get_diag(x, freq=True)
        save(f"g1_linops_{tag}", **out)


def g2_psf2otf():
    from dprox.utils.psf2otf import psf2otf
    out = {}
    k15 = gauss(15, 5.0)
    out["k15"] = k15
    o = psf2otf(k15, [64, 64, 1]); out["g15_64x64x1"] = o; out["g15_64x64x1_isreal"] = np.isrealobj(o)
    o = psf2otf(k15, [32, 48, 3]); out["g15_32x48x3"] = o
    o = psf2otf(k15, [33, 47, 3]); out["g15_33x47x3"] = o
    rng = np.random.RandomState(5)
    ka = rng.rand(5, 4, 1).astype("float32")          # asymmetric, even width
    out["ka"] = ka
    out["ka_24x20x3"] = psf2otf(ka, [24, 20, 3])
    for d in (0, 1, 2):
        D = dp.grad(dp.Variable(), dim=d).kernel
        out[f"D{d}"] = D
        out[f"D{d}_16x24x3"] = psf2otf(D, [16, 24, 3])
    save("g2_psf2otf", **out)


def g3_prox():
    rng = np.random.RandomState(103)
    v = T(rng.randn(2, 3, 16, 24).astype("float32"))
    lam0 = torch.tensor(0.3)
    lamB = torch.tensor([0.3, 0.05])
    var = dp.Variable()
    var.value = torch.zeros(2, 3, 16, 24)
    out = {"v": v, "lamB": lamB}
    from dprox.proxfn.norm import soft_threshold
    out["soft_0p3"] = soft_threshold(v, 0.3)
    out["norm1_scalar"] = dp.norm1(var).prox(v, lam0)
    out["norm1_batch"] = dp.norm1(var).prox(v, lamB)
    out["norm1_alpha2p5"] = (2.5 * dp.norm1(var)).prox(v, lamB)
    c = T(rng.randn(2, 3, 16, 24).astype("float32"))
    out["c"] = c
    out["norm1_offset"] = dp.norm1(var - c).prox(v, lamB)
    out["norm1_grad1_offset"] = dp.norm1(dp.grad(var, dim=1) - c).prox(v, lam0)
    out["nonneg"] = dp.nonneg(var).prox(v, lam0)
    out["nonneg_offset"] = dp.nonneg(var - c).prox(v, lam0)
    out["sumsq_batch"] = dp.sum_squares(var).prox(v, lamB)
    out["norm2_scalar"] = dp.norm2(var).prox(v, lam0)
    fn = dp.norm1(var); fn.beta = 2.0
    out["norm1_beta2"] = fn.prox(v, lam0)
    save("g3_prox", **out)


def _tv_problem(x, b, psf, dims=(0, 1)):
    fns = dp.sum_squares(dp.conv(x, psf) - b)
    for d in dims:
        fns = fns + dp.norm1(dp.grad(x, dim=d))
    return fns


def g4_solve_direct():
    rng = np.random.RandomState(104)
    B, C, H, W = 2, 3, 24, 32
    b = T(rng.rand(B, C, H, W).astype("float32"))
    psf = gauss(9, 2.5)
    x = dp.Variable()
    fns = _tv_problem(x, b, psf)
    solver = dp.compile(fns, method="admm", device="cpu")
    solver.Kall.update_vars([b])
    rhs = [T(rng.randn(B, C, H, W).astype("float32")) for _ in range(2)]
    out = {"b": b, "psf": psf, "rhs0": rhs[0], "rhs1": rhs[1]}
    out["x_rho_scalar"] = solver.least_square.solve(rhs, torch.tensor(0.7))
    out["x_rho_batch"] = solver.least_square.solve(rhs, torch.tensor([0.7, 0.05]))
    # identity Psi term: |H|^2 + rho*1
    x2 = dp.Variable()
    fns2 = dp.sum_squares(dp.conv(x2, psf) - b) + dp.nonneg(x2)
    s2 = dp.compile(fns2, method="admm", device="cpu")
    s2.Kall.update_vars([b])
    out["x_identity"] = s2.least_square.solve([rhs[0]], torch.tensor(0.3))
    save("g4_solve_direct", **out)


def _sample(t):
    return t[..., ::4, ::4]


def g5_admm_tv():
    # --- config 1 exactly: 1x1x256x256, rho 0.1, lam 0.005, 20 iterations
    gt, b, psf = synthetic.deconv_case(1, 1, 256, 256, seed=2023)
    x = dp.Variable()
    fns = _tv_problem(x, T(b), psf)
    snaps = {}

    def cb(iter, state, rho, lam):
        if iter + 1 in (1, 5, 20):
            xs, vs, us = state
            snaps[f"it{iter + 1}_x"] = _sample(xs).clone()
            for i in range(2):
                snaps[f"it{iter + 1}_v{i}"] = _sample(vs[i]).clone()
                snaps[f"it{iter + 1}_u{i}"] = _sample(us[i]).clone()
                snaps[f"it{iter + 1}_v{i}_sum"] = vs[i].double().sum()
                snaps[f"it{iter + 1}_u{i}_l2"] = us[i].double().norm()
            snaps[f"it{iter + 1}_x_sum"] = xs.double().sum()

    out = dp.Problem(fns).solve(method="admm", device="cpu", x0=T(b), rhos=0.1, lams=0.005, max_iter=20, callback=cb)
    psnr = 10 * np.log10(1.0 / np.mean((out.numpy() - gt) ** 2))
    lam20 = np.full(20, 0.005, np.float32)
    x64o, _, _ = admm_f64(b, psf, [("grad0", "norm1", 1.0), ("grad1", "norm1", 1.0)], np.full(20, 0.1, np.float32), [lam20, lam20], 20)
    x64 = ref_f64_admm(b, psf, 20)                          # the reference itself in float64
    f64_pin(snaps, "x_f64", x64, x64o)
    save("g5_admm_tv_c1", gt=gt, b=b, psf=psf, x=out, psnr=psnr, value_after=x.value, x_f64=x64, **snaps)

    # --- 2x3x64x64, 50 iterations, per-iteration rho schedule, full final state
    gt, b, psf = synthetic.deconv_case(2, 3, 64, 64, seed=7)
    x = dp.Variable()
    fns = _tv_problem(x, T(b), psf)
    rhos = torch.linspace(0.05, 0.3, 50)
    st = dp.Problem(fns).solve(method="admm", device="cpu", x0=T(b), rhos=rhos, lams=0.004, max_iter=50,
                               return_full_states=True)
    save("g5_admm_tv_small", gt=gt, b=b, psf=psf, rhos=rhos, x=st[0], v0=st[1][0], v1=st[1][1], u0=st[2][0], u1=st[2][1])

    # --- defaults (rho=1, lam=0.02, 24 it), HWC numpy x0, three grad terms incl. channel dim, per-term lams
    gt, b, psf = synthetic.deconv_case(1, 3, 32, 40, seed=9)
    x = dp.Variable()
    t_data = dp.sum_squares(dp.conv(x, psf) - T(b))
    t0, t1, t2 = dp.norm1(dp.grad(x, dim=0)), 2.0 * dp.norm1(dp.grad(x, dim=1)), dp.norm1(dp.grad(x, dim=2))
    out_def = dp.Problem(t_data + t0 + t1 + t2).solve(method="admm", device="cpu", x0=np.ascontiguousarray(b[0].transpose(1, 2, 0)))
    out_lams = dp.Problem(t_data + t0 + t1 + t2).solve(
        method="admm", device="cpu", x0=T(b), rhos=0.2, max_iter=6,
        lams={t0: 0.01, t1: torch.linspace(0.01, 0.02, 6), t2: 0.003})
    save("g5_admm_tv_misc", b=b, psf=psf, x_defaults=out_def, x_lams=out_lams)


def _csmri(B, H, W, seed):
    gt, mask, y = synthetic.csmri_case(B, H, W, seed=seed, center=8)
    return gt, T(mask), T(y)


def g6_cg():
    out = {}
    for B in (1, 4):
        gt, mask, y = _csmri(B, 32, 32, seed=60 + B)
        rho = 0.35

        class A(torch.nn.Module):
            def forward(self, x):
                return ifft2(mask * (mask * fft2(x))).real + rho * x

        rhs = ifft2(mask * y).real.float() + rho * T(gt)
        iters = []
        import dprox.linalg.solve.solver_cg as scg
        orig = scg.torch.linalg.norm
        xs = ref_cg(A(), rhs, rtol=1e-6, max_iters=100, verbose=False)
        # iteration count at exit: re-run with verbose and parse
        import io, contextlib
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ref_cg(A(), rhs, rtol=1e-6, max_iters=100, verbose=True)
        txt = buf.getvalue()
        n = int(txt.split("Converged at CG Iter")[1].split()[0]) if "Converged" in txt else 100
        out[f"B{B}_mask"], out[f"B{B}_rhs"], out[f"B{B}_x"], out[f"B{B}_iters"] = mask, rhs, xs, n
        xs10 = ref_cg(A(), rhs, rtol=0.0, max_iters=10)
        out[f"B{B}_x_10it"] = xs10
    out["rho"] = 0.35
    save("g6_cg", **out)


def g6b_cg_large_batches():
    """cg() on the config-4 operator for batches beyond 8 systems (12 and 20 x 1 x 32 x 32; per-image rho): solution, exit iteration
    and the iterate after 10 fixed iterations -- the spectral-norm stop rule couples all images of a batch
    (linalg/solve/solver_cg.py:95-129).  Small enough for the SIMT emulator; pins both branches of dpx_cg_masked_fft."""
    import contextlib
    import io
    out = {}
    for B in (12, 20):
        gt, mask, y = _csmri(B, 32, 32, seed=600 + B)
        rho = torch.from_numpy((0.3 + 0.02 * np.arange(B)).astype(np.float32)).view(B, 1, 1, 1)

        class A(torch.nn.Module):
            def forward(self, x):
                return ifft2(mask * (mask * fft2(x))).real + rho * x

        rhs = ifft2(mask * y).real.float() + rho * T(gt)
        xs = ref_cg(A(), rhs, rtol=1e-6, max_iters=100, verbose=False)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ref_cg(A(), rhs, rtol=1e-6, max_iters=100, verbose=True)
        txt = buf.getvalue()
        n = int(txt.split("Converged at CG Iter")[1].split()[0]) if "Converged" in txt else 100
        print(f"g6b B={B}: reference exits at CG iteration {n}")
        out[f"B{B}_mask"], out[f"B{B}_rhs"], out[f"B{B}_rho"], out[f"B{B}_x"], out[f"B{B}_iters"] = mask, rhs, rho.view(B), xs, n
        out[f"B{B}_x_10it"] = ref_cg(A(), rhs, rtol=0.0, max_iters=10)
    save("g6b_cg_large_batches", **out)


def g7_ladmm_cg():
    B, H, W = 2, 32, 32
    gt, mask, y = _csmri(B, H, W, seed=70)
    x = dp.Variable()
    A = MaskedFFT(x, mask)
    den = GrayDen(seed=11)
    fns = dp.sum_squares(A, y) + dp.nonneg(x) + dp.deep_prior(x, denoiser=den)
    x0 = ifft2(y).real.float()
    st = dp.Problem(fns, linear_solve_config=LinearSolveConfig(rtol=1e-6, max_iters=100)).solve(
        method="ladmm", device="cpu", x0=x0, rhos=0.5, lams=0.03, max_iter=5, return_full_states=True)
    st_admm = dp.Problem(fns, linear_solve_config=LinearSolveConfig(rtol=1e-6, max_iters=100)).solve(
        method="admm", device="cpu", x0=x0, rhos=0.5, lams=0.03, max_iter=3)
    save("g7_ladmm_cg", gt=gt, mask=mask, y=y, x0=x0, x=st[0], v0=st[1][0], v1=st[1][1], u0=st[2][0], u1=st[2][1],
         x_admm=st_admm)


def g8_ffdnet():
    rng = np.random.RandomState(108)
    out = {}
    col = ColorDen(7).eval()
    for tag, shape in (("odd", (1, 3, 33, 47)), ("even", (2, 3, 32, 40))):
        x = T(rng.rand(*shape).astype("float32"))
        out[f"{tag}_x"] = x
        for s in (0.02, 0.2):
            with torch.no_grad():
                out[f"{tag}_s{s}"] = col.denoise(x, torch.tensor(s))
    xb = T(rng.rand(2, 3, 16, 24).astype("float32"))
    with torch.no_grad():
        out["batch_sigma_x"] = xb
        out["batch_sigma"] = col.denoise(xb, torch.tensor([0.05, 0.15]))
    gray = GrayDen(11).eval()
    xg = T(rng.rand(2, 2, 20, 26).astype("float32"))
    with torch.no_grad():
        out["gray_x"] = xg
        out["gray_s0.1"] = gray.denoise(xg, torch.tensor(0.1))
    save("g8_ffdnet", **out)


def g8b_ffdnet_wide_range():
    """FFDNet-colour with LARGE-dynamic-range weights (He-normal x 8 instead of x 0.5: activations grow ~8x per layer and leave the
    binary16 range after a few layers; fp32 carries them): the forward the split-f16 -> split-bf16 fallback is pinned on
    (models/network_ffdnet.py:54-68)."""
    rng = np.random.RandomState(1808)
    col = ColorDen(7)
    col.model = load_ffdnet(3, 3, 96, 12, 7, gain=8.0)
    col = col.eval()
    x = T(rng.rand(1, 3, 32, 40).astype("float32"))
    with torch.no_grad():
        y = col.denoise(x, torch.tensor(0.05))
    assert torch.isfinite(y).all() and float(y.abs().max()) > 1e6
    save("g8b_ffdnet_wide_range", x=x, y=y, gain=np.float32(8.0), sigma=np.float32(0.05))


def g9_admm_pnp():
    gt, b, psf = synthetic.deconv_case(2, 3, 32, 40, seed=90)
    x = dp.Variable()
    prior = dp.deep_prior(x, denoiser=ColorDen(7))
    fns = dp.sum_squares(dp.conv(x, psf) - T(b)) + prior
    rhos, sigmas = log_descent(35, 5, 3)
    st = dp.Problem(fns).solve(method="admm", device="cpu", x0=T(b), rhos=rhos, lams={prior: sigmas}, max_iter=3, return_full_states=True)
    # + nonneg, like the reference's test_algorithms
    x2 = dp.Variable()
    prior2, nn2 = dp.deep_prior(x2, denoiser=ColorDen(7)), dp.nonneg(x2)
    fns2 = dp.sum_squares(dp.conv(x2, psf) - T(b)) + prior2 + nn2
    out2 = dp.Problem(fns2).solve(method="admm", device="cpu", x0=T(b), rhos=rhos, lams={prior2: sigmas, nn2: 0.0}, max_iter=3)
    # the exact (float64) iterates of the same algorithm: context for the fp32 round-off of ill-conditioned x-updates
    x64o, v64o, u64 = admm_f64(b, psf, [("id", "ffdnet", 1.0)], rhos.numpy(), [sigmas.numpy()], 3, ffdnet_weights(7))
    x64no, _, _ = admm_f64(b, psf, [("id", "ffdnet", 1.0), ("id", "nonneg", 1.0)], rhos.numpy(), [sigmas.numpy(), np.zeros(3)], 3,
                           ffdnet_weights(7))
    # ... produced by the reference itself in float64 (its own FFDNet module cast to double)
    st64 = ref_f64_admm(b, psf, 3, rhos=rhos, lams=sigmas, dims=(), prior=ColorDen(7), full=True)
    x64n = ref_f64_admm(b, psf, 3, rhos=rhos, lams=sigmas, dims=(), prior=ColorDen(7), extra=dp.nonneg)
    pins = {}
    f64_pin(pins, "x_f64", st64[0], x64o, tol=1e-10)
    f64_pin(pins, "v0_f64", st64[1][0], v64o[0], tol=1e-10)
    f64_pin(pins, "x_nonneg_f64", x64n, x64no, tol=1e-10)
    save("g9_admm_pnp", gt=gt, b=b, psf=psf, rhos=rhos, sigmas=sigmas, x=st[0], v0=st[1][0], u0=st[2][0], x_nonneg=out2,
         x_f64=st64[0], v0_f64=st64[1][0], x_nonneg_f64=x64n, **pins)


def g23_pnp_scaled_sqrt():
    """`c * deep_prior(x, sqrt=True)` with c != 1: the noise level is sqrt(c * lam) (ProxFn.prox scales lam by alpha before
    deep_prior._prox takes the root, proxfn/base.py:55-64, pnp/prior.py:77); well-conditioned rho so that 1e-5 parity is meaningful."""
    gt, b, psf = synthetic.deconv_case(2, 3, 32, 40, seed=230)
    x = dp.Variable()
    prior = 0.6 * dp.deep_prior(x, denoiser=ColorDen(7), sqrt=True)
    fns = dp.sum_squares(dp.conv(x, psf) - T(b)) + prior
    rhos = torch.tensor([0.5, 0.4, 0.3])
    lams = torch.tensor([0.004, 0.003, 0.002])
    with torch.no_grad():
        st = dp.Problem(fns).solve(method="admm", device="cpu", x0=T(b), rhos=rhos, lams={prior: lams}, max_iter=3, return_full_states=True)
    save("g23_pnp_scaled_sqrt", b=b, psf=psf, rhos=rhos, lams=lams, x=st[0], v0=st[1][0], u0=st[2][0])


def g24_linear_solve_grad():
    """linear_solve with the implicit backward pass (linalg/custom.py:39-82) on the masked-Fourier normal operator
    A_rho(x) = Re F^H M F x + rho x with a trainable per-image rho: x, dL/db (one transposed solve) and dL/drho (operator VJP at
    the solution) for L = <x, w>; the reference's own tests of this rule: tests/linalg/test_linear_solver_grad.py:101-123,
    tests/linalg/test_linear_solver_torch.py:51-95."""
    from dprox.linalg import linear_solve
    B, H, W = 2, 32, 32
    gt, mask, y = _csmri(B, H, W, seed=240)
    rng = np.random.RandomState(241)

    class Normal(torch.nn.Module):
        def __init__(self, rho):
            super().__init__()
            self.rho = torch.nn.Parameter(rho)

        def forward(self, x):
            return ifft2(mask * (mask * fft2(x))).real.float() + self.rho.view(-1, 1, 1, 1) * x

        @property
        def T(self):
            return self

        def clone(self):
            return Normal(self.rho.detach().clone())

    rho0 = torch.tensor([0.35, 0.6])
    A = Normal(rho0.clone())
    b = (ifft2(mask * y).real.float() + rho0.view(-1, 1, 1, 1) * T(gt)).clone().requires_grad_(True)
    w = T(rng.randn(B, 1, H, W).astype("float32"))
    x = linear_solve(A, b, LinearSolveConfig(rtol=1e-6, max_iters=100))
    (x * w).sum().backward()
    save("g24_linear_solve_grad", mask=mask, rho=rho0, b=b.detach(), w=w, x=x.detach(), g_b=b.grad, g_rho=A.rho.grad)
    print("g24", A.rho.grad, float(b.grad.abs().max()))


def g21_x8_augment():
    """deep_prior(x8=True): denoisers/composite.py:6-46 -- nine consecutive prox calls (modes 0..7, 0) on a non-square image"""
    x = dp.Variable()
    prior = dp.deep_prior(x, denoiser=ColorDen(7), x8=True)
    v = T(np.random.RandomState(210).rand(2, 3, 26, 38).astype("float32"))
    outs = []
    with torch.no_grad():
        for k in range(9):
            outs.append(prior._prox(v, torch.tensor(0.02 + 0.01 * k)))
    save("g21_x8_augment", v=v, outs=torch.stack(outs))


def g11_unrolled_grads():
    """Config-5 shape of problem at fixture size: unrolled ADMM (specialize method='unroll', shared solver), MSE loss,
    gradients w.r.t. the per-iteration rho / lambda schedules and the observation (README.md:93-116,
    specialization/unroll.py:14-18)."""
    gt, b, psf = synthetic.deconv_case(2, 3, 32, 40, seed=110)
    K = 3
    out = {"gt": gt, "b": b, "psf": psf}
    for tag, with_nonneg in (("tv", False), ("tvnn", True)):
        x = dp.Variable()
        bt = T(b).clone().requires_grad_(True)
        n0, n1 = dp.norm1(dp.grad(x, dim=0)), dp.norm1(dp.grad(x, dim=1))
        fns = dp.sum_squares(dp.conv(x, psf) - bt) + n0 + n1
        if with_nonneg:
            nn_ = dp.nonneg(x)
            fns = fns + nn_
        solver = dp.compile(fns, method="admm", device="cpu")
        solver = dp.specialize(solver, method="unroll", device="cpu", max_iter=K)
        rhos = torch.tensor([0.3, 0.2, 0.1], requires_grad=True)
        l0 = torch.tensor([0.02, 0.015, 0.01], requires_grad=True)
        l1 = torch.tensor([0.03, 0.02, 0.012], requires_grad=True)
        lams = {n0: l0, n1: l1}
        if with_nonneg:
            lams[nn_] = torch.zeros(K)
        x0 = T(b).clone().requires_grad_(True)
        xo = solver.solve(x0=x0, rhos=rhos, lams=lams)
        loss = ((xo - T(gt)) ** 2).mean()
        loss.backward()
        out.update({f"{tag}_x": xo, f"{tag}_loss": loss, f"{tag}_g_rhos": rhos.grad, f"{tag}_g_l0": l0.grad, f"{tag}_g_l1": l1.grad,
                    f"{tag}_g_b": bt.grad, f"{tag}_g_x0": x0.grad if x0.grad is not None else torch.zeros_like(x0)})
        print(tag, float(loss), rhos.grad, l0.grad, l1.grad, float(bt.grad.abs().max()), None if x0.grad is None else float(x0.grad.abs().max()))
    # UnrolledSolver proper (share=False: one solver clone per step, learned rho / lambda parameters); the reference's
    # per-step lam dict is keyed by psi_fns[0] only (unroll.py:54), i.e. it serves single-Psi problems
    # (norm1(x): |H|^2 + rho >= rho keeps the x-update well conditioned; a single grad term leaves a whole line of
    #  frequencies with denominator ~eps, where fp32 round-off, not the algorithm, decides the output)
    x = dp.Variable()
    n1 = dp.norm1(x)
    solver = dp.compile(dp.sum_squares(dp.conv(x, psf) - T(b)) + n1, method="admm", device="cpu")
    us = dp.specialize(solver, method="unroll", device="cpu", max_iter=K, share=False, learned_params=True)
    with torch.no_grad():
        us.rhos.copy_(torch.tensor([0.3, 0.2, 0.1]))
        list(us.lams.values())[0].copy_(torch.tensor([0.03, 0.02, 0.012]))
    xo = us.solve(x0=T(b))
    loss = ((xo - T(gt)) ** 2).mean()
    loss.backward()
    out.update(us_x=xo, us_loss=loss, us_g_rhos=us.rhos.grad, us_g_lam=list(us.lams.values())[0].grad)
    print("us", float(loss.detach()), us.rhos.grad, list(us.lams.values())[0].grad)
    out["rhos"], out["l0"], out["l1"] = np.array([0.3, 0.2, 0.1], "float32"), np.array([0.02, 0.015, 0.01], "float32"), np.array([0.03, 0.02, 0.012], "float32")
    save("g11_unrolled_grads", **out)


def g16_ffdnet_grads():
    """Backward through the FFDNet stack (PyTorch autograd of the reference's network_ffdnet.py:54-68): gradients w.r.t.
    the image and the per-image noise level, odd sizes included (adjoint of the replicate padding); and 2 unrolled
    plug-and-play ADMM iterations with gradients w.r.t. rho_t, sigma_t and x0."""
    rng = np.random.RandomState(160)
    out = {}
    col = ColorDen(7).eval()
    for p in col.parameters():
        p.requires_grad_(False)
    for tag, shape in (("odd", (2, 3, 33, 47)), ("even", (1, 3, 32, 40))):
        x = T(rng.rand(*shape).astype("float32")).requires_grad_(True)
        sig = torch.tensor([0.05, 0.2][: shape[0]], requires_grad=True)
        w = T(rng.randn(*shape).astype("float32"))
        y = col.denoise(x, sig)
        (y * w).sum().backward()
        out.update({f"{tag}_x": x.detach(), f"{tag}_sigma": sig.detach(), f"{tag}_w": w, f"{tag}_y": y.detach(), f"{tag}_gx": x.grad, f"{tag}_gsigma": sig.grad})
    gray = GrayDen(11).eval()
    for p in gray.parameters():
        p.requires_grad_(False)
    xg = T(rng.rand(2, 2, 21, 26).astype("float32")).requires_grad_(True)
    sg = torch.tensor(0.1, requires_grad=True)
    wg = T(rng.randn(2, 2, 21, 26).astype("float32"))
    yg = gray.denoise(xg, sg)
    (yg * wg).sum().backward()
    out.update(gray_x=xg.detach(), gray_w=wg, gray_y=yg.detach(), gray_gx=xg.grad, gray_gsigma=sg.grad)
    # weight / bias gradients (trainable prior): first, one middle and last layer of the colour net, all of the gray net's
    colt = ColorDen(7).train()
    xw = T(rng.rand(2, 3, 24, 38).astype("float32"))
    ww = T(rng.randn(2, 3, 24, 38).astype("float32"))
    (colt.denoise(xw, torch.tensor([0.05, 0.2])) * ww).sum().backward()
    convs = [m for m in colt.model.model if isinstance(m, torch.nn.Conv2d)]
    out.update(wg_x=xw, wg_w=ww)
    for li in (0, 5, 11):
        out[f"wg_dw{li}"], out[f"wg_db{li}"] = convs[li].weight.grad, convs[li].bias.grad
    # unrolled PnP
    gt, b, psf = synthetic.deconv_case(2, 3, 32, 40, seed=161)
    x = dp.Variable()
    prior = dp.deep_prior(x, denoiser=ColorDen(7))
    solver = dp.compile(dp.sum_squares(dp.conv(x, psf) - T(b)) + prior, method="admm", device="cpu")
    solver = dp.specialize(solver, method="unroll", device="cpu", max_iter=2)
    rhos = torch.tensor([0.4, 0.2], requires_grad=True)
    sigmas = torch.tensor([0.08, 0.04], requires_grad=True)
    x0 = T(b).clone().requires_grad_(True)
    xo = solver.solve(x0=x0, rhos=rhos, lams={prior: sigmas})
    loss = ((xo - T(gt)) ** 2).mean()
    loss.backward()
    out.update(pnp_gt=gt, pnp_b=b, pnp_psf=psf, pnp_x=xo.detach(), pnp_loss=loss.detach(), pnp_g_rhos=rhos.grad, pnp_g_sigmas=sigmas.grad,
               pnp_g_x0=x0.grad)
    print("pnp", float(loss.detach()), rhos.grad, sigmas.grad, float(x0.grad.abs().max()))
    save("g16_ffdnet_grads", **out)


def g17_mosaic_jd():
    """mosaic / mul_elementwise linops and the joint demosaic + deconvolution problem of tests/problem/test_jd23.py
    (sum_squares(mosaic(conv(x, psf)) - b) + deep_prior, ADMM with the CG x-update) at fixture size."""
    from dprox.contrib import mosaicing
    rng = np.random.RandomState(170)
    out = {}
    x = T(rng.rand(2, 3, 12, 14).astype("float32"))
    m = dp.mosaic(dp.Variable())
    out.update(lin_x=x, mosaic_fwd=m.forward(x), mosaic_adj=m.adjoint(x), mosaic_diag=m.get_diag(x).detach())
    w = rng.rand(1, 3, 12, 14).astype("float32")
    me = dp.mul_elementwise(dp.Variable(), w)
    out.update(mul_w=w, mul_fwd=me.forward(x), mul_adj=me.adjoint(x))
    P = dp.Placeholder()
    mc = dp.mul_color(dp.Variable(), P)                      # the SRF must be a 2-D tensor (mul.py:36-42 uses srf.T)
    srf = T(np.random.RandomState(172).rand(3, 5).astype("float32"))
    P.value = srf
    x5 = T(np.random.RandomState(173).rand(2, 5, 12, 14).astype("float32"))
    out.update(srf=srf, mulc_fwd=mc.forward(x).detach(), mulc_x5=x5, mulc_adj=mc.adjoint(x5).detach())
    # weighted_sum_squares with a mul_elementwise weight (sum_square.py:51-75): a Psi-side prox
    bw = T(np.random.RandomState(174).rand(2, 3, 12, 14).astype("float32"))
    wss = dp.weighted_sum_squares(dp.Variable(), dp.mul_elementwise(dp.Variable(), w), bw)
    out.update(wss_b=bw, wss_prox=wss.prox(x, torch.tensor([0.3, 1.2])).detach())
    gt, blur, psf = synthetic.deconv_case(2, 3, 32, 40, seed=171)
    b = mosaicing(T(blur[0].transpose(1, 2, 0)))            # the reference helper takes one HWC image
    b = torch.cat([b, mosaicing(T(blur[1].transpose(1, 2, 0)))], dim=0).float()
    xv = dp.Variable()
    data = dp.sum_squares(dp.mosaic(dp.conv(xv, psf)) - b)
    reg = dp.deep_prior(xv, denoiser=ColorDen(7))
    prob = dp.Problem(data + reg, linear_solve_config=LinearSolveConfig(max_iters=50))
    # rho large enough for CG to converge (rtol 1e-6) well inside its 50 iterations: with the test's log_descent(35, 30)
    # schedule (rho ~ 1e-4) the x-update is the 50th iterate of an unconverged CG, which amplifies fp32 round-off to 1e-3
    _, sigmas = log_descent(35, 30, 3)
    rhos = torch.tensor([0.5, 0.4, 0.3])
    with torch.no_grad():
        st = prob.solve(method="admm", device="cpu", x0=b, rhos=rhos, lams={reg: sigmas}, max_iter=3, return_full_states=True)
    out.update(jd_b=b, jd_psf=psf, jd_rhos=rhos, jd_sigmas=sigmas, jd_x=st[0], jd_v=st[1][0], jd_u=st[2][0])
    save("g17_mosaic_jd", **out)


def g18_sisr():
    """sisr closed-form data term (proxfn/fast/sr.py:45-77): the prox alone (scalar and per-image lam, sf = 2 and 3) and the
    super-resolution example (examples/applications/super_resolution.py) with the FFDNet prior, 3 ADMM iterations."""
    import scipy.ndimage
    rng = np.random.RandomState(180)
    out = {}
    psf = synthetic.point_spread_function(5, 3.0)                       # HxWx1
    for sf, (h, w) in ((2, (12, 10)), (3, (8, 9))):
        gt = synthetic.synth(rng, 2, 3, h * sf, w * sf)
        blur = np.stack([np.stack([scipy.ndimage.convolve(gt[b, c], psf[..., 0], mode="wrap") for c in range(3)]) for b in range(2)])
        y = blur[..., ::sf, ::sf].astype("float32")
        x = dp.Variable()
        fn = dp.sisr(x, T(y), kernel=psf, sf=sf)
        v = T(rng.rand(2, 3, h * sf, w * sf).astype("float32"))
        out.update({f"sf{sf}_y": y, f"sf{sf}_v": v, f"sf{sf}_prox_scalar": fn._prox(v, torch.tensor(0.4), 1),
                    f"sf{sf}_prox_B": fn._prox(v, torch.tensor([0.2, 0.9]).view(2, 1, 1, 1), 2)})
    out["psf"] = psf
    # the example problem
    gt = synthetic.synth(rng, 1, 3, 32, 40)
    blur = np.stack([scipy.ndimage.convolve(gt[0, c], psf[..., 0], mode="wrap") for c in range(3)])[None]
    y = blur[..., ::2, ::2].astype("float32")
    x0 = np.repeat(np.repeat(y, 2, axis=-2), 2, axis=-1)                  # nearest-neighbour start (the example uses cv2 bicubic)
    x = dp.Variable()
    data = dp.sisr(x, T(y), kernel=psf, sf=2)
    reg = dp.deep_prior(x, denoiser=ColorDen(7))
    prob = dp.Problem(data + reg)
    # (rho ~ 1e-4 of the example's log_descent(35, 35) schedule divides the spectrum by I rho: fp32 round-off of either
    #  implementation is amplified 1e4x; the fixture keeps the x-update well conditioned)
    _, sigmas = log_descent(35, 35, 3)
    rhos = torch.tensor([0.6, 0.4, 0.3])
    with torch.no_grad():
        st = prob.solve(method="admm", device="cpu", x0=T(x0), rhos=rhos, lams={reg: sigmas}, max_iter=3, return_full_states=True)
    out.update(sr_y=y, sr_x0=x0, sr_rhos=rhos, sr_sigmas=sigmas, sr_x=st[0], sr_v=st[1][0], sr_u=st[2][0])
    save("g18_sisr", **out)


def g19_conv_doe():
    """conv_doe (linop/conv.py:81-156): PSF given as a tensor / Placeholder, OTF rebuilt per value through psf2otf2 (incl. its
    padding split and the all-dims ifftshift); forward / adjoint / get_diag and an ADMM TV solve with the PSF in a Placeholder."""
    from dprox.linop.conv import conv_doe
    rng = np.random.RandomState(190)
    out = {}
    for tag, (C, H, W, f) in (("odd", (3, 24, 24, 9)), ("even", (1, 20, 24, 8))):
        psf = rng.rand(1, C, f, f + (W - H)).astype("float32")
        psf /= psf.sum(axis=(-2, -1), keepdims=True)
        x = T(rng.rand(2, C, H, W).astype("float32"))
        op = conv_doe(dp.Variable(), T(psf))
        with torch.no_grad():
            out.update({f"{tag}_psf": psf, f"{tag}_x": x, f"{tag}_fwd": op.forward(x), f"{tag}_adj": op.adjoint(x), f"{tag}_diag": op.get_diag(x, freq=True)})
    gt, _, _ = synthetic.deconv_case(2, 3, 32, 32, seed=191)
    psf = rng.rand(1, 3, 7, 7).astype("float32") ** 3
    psf /= psf.sum(axis=(-2, -1), keepdims=True)
    xv = dp.Variable()
    P, Y = dp.Placeholder(), dp.Placeholder()
    opb = conv_doe(dp.Variable(), T(psf))
    with torch.no_grad():
        y = opb.forward(T(gt)) + T((rng.randn(2, 3, 32, 32) * 0.01).astype("float32"))
    n0, n1 = dp.norm1(dp.grad(xv, dim=0)), dp.norm1(dp.grad(xv, dim=1))
    fns = dp.sum_squares(conv_doe(xv, P, circular=True), Y) + n0 + n1     # conv_doe registers its watcher on P here
    P.value, Y.value = T(psf), y            # ... the values must exist before compile (CompGraph reads them)
    solver = dp.compile(fns, method="admm", device="cpu")
    with torch.no_grad():
        st = solver.solve(x0=y, rhos=0.2, lams=0.01, max_iter=8, return_full_states=True)
    out.update(tv_psf=psf, tv_y=y, tv_x=st[0], tv_v0=st[1][0], tv_u0=st[2][0])
    # circular=False: zero-pad to 2H x 2H, circular product, crop (conv.py:100-108); + an ADMM TV solve through it
    psfl = rng.rand(1, 3, 7, 7).astype("float32") ** 2
    psfl /= psfl.sum(axis=(-2, -1), keepdims=True)
    xl = T(rng.rand(2, 3, 24, 24).astype("float32"))
    opl = conv_doe(dp.Variable(), T(psfl), circular=False)
    with torch.no_grad():
        out.update(lin_psf=psfl, lin_x=xl, lin_fwd=opl.forward(xl), lin_adj=opl.adjoint(xl))
    xv2 = dp.Variable()
    with torch.no_grad():
        yl = opl.forward(T(gt[:, :, :24, :24].copy())) + T((rng.randn(2, 3, 24, 24) * 0.01).astype("float32"))
    fl = dp.sum_squares(conv_doe(xv2, T(psfl), circular=False), yl) + dp.norm1(dp.grad(xv2, dim=0)) + dp.norm1(dp.grad(xv2, dim=1))
    with torch.no_grad():
        xs = dp.compile(fl, method="admm", device="cpu").solve(x0=yl, rhos=0.3, lams=0.01, max_iter=6)
    out.update(lin_y=yl, lin_tv_x=xs)
    save("g19_conv_doe", **out)


def g25_doe_psf_grad():
    """End-to-end optics (README.md:93-116): the PSF of a conv_doe data term sits in a Placeholder, the solver is unrolled, and the
    loss is differentiated w.r.t. the PSF, the observation and the schedules by the reference's autograd (linop/conv.py:81-156:
    the Placeholder's value is wrapped in an nn.Parameter whose .grad receives the PSF gradient)."""
    from dprox.linop.conv import conv_doe
    rng = np.random.RandomState(250)
    out = {}
    for tag, (B, C, H, W, f, K) in (("a", (2, 3, 32, 32, 7, 4)), ("b", (1, 1, 20, 24, 5, 3))):
        gt = rng.rand(B, C, H, W).astype("float32")
        psf = rng.rand(1, C, f, f + (W - H)).astype("float32") ** 2       # (psf2otf2 pads both axes by the height difference)
        psf /= psf.sum(axis=(-2, -1), keepdims=True)
        xv = dp.Variable()
        P, Y = dp.Placeholder(), dp.Placeholder()
        op = conv_doe(xv, P, circular=True)
        with torch.no_grad():
            y = conv_doe(dp.Variable(), T(psf)).forward(T(gt)) + T((rng.randn(B, C, H, W) * 0.01).astype("float32"))
        n0, n1 = dp.norm1(dp.grad(xv, dim=0)), dp.norm1(dp.grad(xv, dim=1))
        yt = y.clone().requires_grad_(True)
        P.value, Y.value = T(psf), yt
        solver = dp.compile(dp.sum_squares(op, Y) + n0 + n1, method="admm", device="cpu")
        solver = dp.specialize(solver, method="unroll", device="cpu", max_iter=K)
        rhos = torch.full((K,), 0.2, requires_grad=True)
        lam = torch.full((K,), 0.01, requires_grad=True)
        xo = solver.solve(x0=y, rhos=rhos, lams={n0: lam, n1: lam})
        loss = ((xo - T(gt)) ** 2).mean()
        loss.backward()
        assert op.psf.grad is not None and yt.grad is not None
        out.update({f"{tag}_gt": gt, f"{tag}_psf": psf, f"{tag}_y": y, f"{tag}_x": xo.detach(), f"{tag}_loss": loss.detach().double(),
                    f"{tag}_g_psf": op.psf.grad, f"{tag}_g_y": yt.grad, f"{tag}_g_rhos": rhos.grad, f"{tag}_g_lam": lam.grad, f"{tag}_K": K})
        print("g25", tag, float(loss), float(op.psf.grad.abs().max()), float(yt.grad.abs().max()))
    save("g25_doe_psf_grad", **out)


def g20_drunet():
    """DRUNet (UNetRes, models/network_unet.py:67-117) behind DRUNetDenoiser (wrapper.py:89-146) with seeded weights:
    the padded single-pass path (<= 256x256, size not a multiple of 16) and the four-quadrant path (> 256x256)."""
    from oracle.dprox_oracle import drunet_weights
    from dprox.proxfn.pnp.denoisers.models.network_unet import UNetRes
    from dprox.proxfn.pnp.denoisers.wrapper import DRUNetDenoiser
    rng = np.random.RandomState(200)
    out = {}
    for tag, n_ch, seed in (("color", 3, 21), ("gray", 1, 22)):
        net = UNetRes(in_nc=n_ch + 1, out_nc=n_ch, nc=[64, 128, 256, 512], nb=4, act_mode="R", downsample_mode="strideconv",
                      upsample_mode="convtranspose")
        net.load_state_dict(drunet_weights(seed, n_ch + 1, n_ch), strict=True)
        den = DRUNetDenoiser.__new__(DRUNetDenoiser)
        Denoiser.__init__(den)
        den.model = net.eval()
        shapes = ((2, n_ch, 40, 52),) if tag == "color" else ((1, n_ch, 33, 47), (1, n_ch, 264, 260))
        for i, shape in enumerate(shapes):
            big = shape[-1] > 256
            x = T((np.random.RandomState(201).rand(*shape) if big else rng.rand(*shape)).astype("float32"))
            sig = torch.tensor([0.05, 0.2][: shape[0]])
            with torch.no_grad():
                y = den.denoise(x, sig)
            if not big:                                       # the large input is regenerated from its seed (201) by the tests
                out[f"{tag}{i}_x"] = x
            out[f"{tag}{i}_y"], out[f"{tag}{i}_sigma"] = y, sig
    # gradients w.r.t. the image and sigma through the colour net (frozen weights), padded single-pass path
    net = UNetRes(in_nc=4, out_nc=3, nc=[64, 128, 256, 512], nb=4, act_mode="R", downsample_mode="strideconv", upsample_mode="convtranspose")
    net.load_state_dict(drunet_weights(21, 4, 3), strict=True)
    net.requires_grad_(False)
    den = DRUNetDenoiser.__new__(DRUNetDenoiser)
    Denoiser.__init__(den)
    den.model = net.eval()
    xg = T(np.random.RandomState(202).rand(2, 3, 24, 40).astype("float32")).requires_grad_(True)
    sg = torch.tensor([0.05, 0.2], requires_grad=True)
    wg = T(np.random.RandomState(203).randn(2, 3, 24, 40).astype("float32"))
    (den.denoise(xg, sg) * wg).sum().backward()
    out.update(grad_x=xg.detach(), grad_w=wg, grad_gx=xg.grad, grad_gsigma=sg.grad)
    # weight gradients of the same loss (trainable denoiser under `unroll`): the two small layers in full, 8x8 corners of one
    # layer of every kind (ResBlock conv at each level, strided 2x2, transposed 2x2) and the L2 norm of every parameter's gradient
    net.requires_grad_(True)
    (den.denoise(xg.detach(), sg.detach()) * wg).sum().backward()
    sd_grads = {n: p.grad for n, p in net.named_parameters()}
    out["wgrad_names"] = np.array(sorted(sd_grads))
    out["wgrad_norms"] = np.array([float(sd_grads[n].norm()) for n in sorted(sd_grads)], dtype=np.float64)
    for n in ("m_head.weight", "m_tail.weight"):
        out["wgrad_full_" + n] = sd_grads[n]
    for n in ("m_down1.0.res.0.weight", "m_down2.4.weight", "m_down3.2.res.2.weight", "m_body.1.res.2.weight", "m_up3.0.weight",
              "m_up2.3.res.0.weight", "m_up1.0.weight"):
        out["wgrad_corner_" + n] = sd_grads[n][:8, :8].contiguous()
    net.requires_grad_(False)
    # IRCNN (dilated convolutions) behind IRCNNDenoiser: two noise-level bins, two bands
    from synthetic import ircnn_weights
    from dprox.proxfn.pnp.denoisers.wrapper import IRCNNDenoiser
    from dprox.proxfn.pnp.denoisers.models.network_dncnn import IRCNN
    ird = IRCNNDenoiser.__new__(IRCNNDenoiser)
    Denoiser2D.__init__(ird)
    ird.model = IRCNN(in_nc=1, out_nc=1, nc=64)
    ird.model25 = {str(k): ircnn_weights(31 + k) for k in (3, 12)}
    ird.former_idx = -1
    xi = T(np.random.RandomState(204).rand(2, 2, 29, 37).astype("float32"))
    with torch.no_grad():
        out.update(ircnn_x=xi, ircnn_y3=ird.denoise(xi, torch.tensor(8 / 255.0)), ircnn_y12=ird.denoise(xi, torch.tensor(25.5 / 255.0)))
    # IRCNN gradients (bin 3): image, and every weight / bias (full for the small first / last layers, 8x8 corner of the dilation-4 layer)
    xig = xi.clone().requires_grad_(True)
    wi = T(np.random.RandomState(205).randn(2, 2, 29, 37).astype("float32"))
    ird.denoise(xig, torch.tensor(8 / 255.0))                      # loads the bin-3 weights
    ird.model.requires_grad_(True)
    (ird.denoise(xig, torch.tensor(8 / 255.0)) * wi).sum().backward()
    gi = {n: p.grad for n, p in ird.model.named_parameters()}
    out.update(ircnn_gw=wi, ircnn_gx=xig.grad, ircnn_gnames=np.array(sorted(gi)),
               ircnn_gnorms=np.array([float(gi[n].norm()) for n in sorted(gi)], dtype=np.float64),
               ircnn_g_w0=gi["model.0.weight"], ircnn_g_b0=gi["model.0.bias"], ircnn_g_w6_corner=gi["model.6.weight"][:8, :8].contiguous(),
               ircnn_g_b6=gi["model.6.bias"], ircnn_g_w12=gi["model.12.weight"], ircnn_g_b12=gi["model.12.bias"])
    save("g20_drunet", **out)


def g22_unet():
    """UNetDenoiser (wrapper.py:206-221 -> models/unet/unet.py:34-135) with seeded weights: forward on an odd-sized two-band image
    (MaxPool floors, the up path zero-pads back: 37x45 -> 18x22 -> 9x11 -> 4x5 -> 2x2) and on an even one with per-image sigma;
    gradients w.r.t. the image and sigma (backward-data pass) through the reference's autograd."""
    from synthetic import unet_weights
    from dprox.proxfn.pnp.denoisers.models.unet import UNet
    from dprox.proxfn.pnp.denoisers.wrapper import UNetDenoiser
    net = UNet(2, 1)
    net.load_state_dict(unet_weights(41), strict=True)
    den = UNetDenoiser.__new__(UNetDenoiser)
    Denoiser2D.__init__(den)
    den.model = net.eval()
    rng = np.random.RandomState(220)
    out = {}
    x_odd = T(rng.rand(1, 2, 37, 45).astype("float32"))
    x_even = T(rng.rand(2, 1, 32, 48).astype("float32"))
    with torch.no_grad():
        out.update(odd_x=x_odd, odd_y=den.denoise(x_odd, torch.tensor(0.1)),
                   even_x=x_even, even_sigma=np.array([0.05, 0.2], "float32"), even_y=den.denoise(x_even, torch.tensor([0.05, 0.2])))
        # the raw network (no clamp) on the even input: pins the residual path separately from the clamp
        nm = torch.ones_like(x_even) * torch.tensor([0.05, 0.2]).view(-1, 1, 1, 1)
        out["even_raw"] = net(torch.cat([x_even, nm], dim=1))
    net.requires_grad_(False)
    xg = T(rng.rand(2, 1, 24, 40).astype("float32") * 0.6 + 0.2).requires_grad_(True)
    sg = torch.tensor([0.05, 0.2], requires_grad=True)
    wg = T(rng.randn(2, 1, 24, 40).astype("float32"))
    (den.denoise(xg, sg) * wg).sum().backward()
    out.update(grad_x=xg.detach(), grad_w=wg, grad_gx=xg.grad, grad_gsigma=sg.grad)
    # weight gradients of the same loss: norms of all, the first / last layers in full
    net.requires_grad_(True)
    (den.denoise(xg.detach(), sg.detach()) * wg).sum().backward()
    gr = {n: p.grad for n, p in net.named_parameters()}
    out["wgrad_names"] = np.array(sorted(gr))
    out["wgrad_norms"] = np.array([float(gr[n].norm()) for n in sorted(gr)], dtype=np.float64)
    for n in ("inc.conv.conv-0.conv2d.weight", "inc.conv.conv-0.conv2d.bias", "outc.conv.weight", "outc.conv.bias", "up4.conv.conv-2.conv2d.bias"):
        out["wgrad_full_" + n] = gr[n]
    out["wgrad_corner_down4"] = gr["down4.mpconv.1.conv-1.conv2d.weight"][:8, :8].contiguous()
    out["wgrad_corner_up1"] = gr["up1.conv.conv-0.conv2d.weight"][:8, :8].contiguous()
    save("g22_unet", **out)


def g15_csmri():
    """CS-MRI pipeline of the reference's examples (csmri closed-form data term + CustomADMM + gray FFDNet prior):
    dprox/proxfn/fast/csmri.py:8-25, dprox/contrib/csmri.py:156-171, ext_sum_squares routing invert.py:8-12."""
    from dprox.contrib.csmri import CustomADMM
    from dprox.proxfn.fast.csmri import csmri
    parts = [synthetic.csmri_case(1, 32, 40, seed=150 + i, center=8) for i in range(2)]   # per-image masks, like the datasets
    gt, mask, y = (np.concatenate([p[k] for p in parts], axis=0) for k in range(3))
    rng = np.random.RandomState(151)
    out = {"gt": gt, "mask": mask, "y": y}
    # (1) the closed-form prox on its own: complex and real inputs, scalar and per-image lam
    x = dp.Variable()
    yp, mp = dp.Placeholder(), dp.Placeholder()
    yp.value, mp.value = T(y), T(mask)
    fn = csmri(x, mp, yp)
    vc = (rng.randn(2, 1, 32, 40) + 1j * rng.randn(2, 1, 32, 40)).astype("complex64")
    out["prox_v"] = vc
    out["prox_lam_scalar"] = fn._prox(T(vc), torch.tensor(0.7), 1)
    out["prox_lam_B"] = fn._prox(T(vc), torch.tensor([0.3, 1.9]), 2)
    # (2) the full solver, 4 iterations
    x2 = dp.Variable()
    y2, m2 = dp.Placeholder(), dp.Placeholder()
    data = csmri(x2, m2, y2)
    reg = dp.deep_prior(x2, denoiser=GrayDen(11))
    solver = CustomADMM([reg], [data])
    y2.value, m2.value = T(y), T(mask)
    x0 = ifft2(T(y))
    rhos, sigmas = log_descent(80, 40, 4)
    rhos, _ = log_descent(10, 0.1, 4)
    with torch.no_grad():
        st = solver.solve(x0=x0, rhos=rhos, lams={reg: sigmas}, max_iter=4, return_full_states=True)
    out.update(x0=x0, rhos=rhos, sigmas=sigmas, x=st[0], z=st[1][0], u=st[2][0])
    save("g15_csmri", **out)


def g10_pgd():
    gt, b, psf = synthetic.deconv_case(2, 3, 32, 40, seed=100)
    x = dp.Variable()
    fns = dp.sum_squares(dp.conv(x, psf) - T(b)) + dp.norm1(x)
    out = dp.Problem(fns).solve(method="pgd", device="cpu", x0=T(b), rhos=0.8, lams=0.01, max_iter=5)
    x2 = dp.Variable()
    fns2 = dp.sum_squares(dp.conv(x2, psf) - T(b)) + dp.nonneg(x2)
    out2 = dp.Problem(fns2).solve(method="pgd", device="cpu", x0=T(b), rhos=torch.tensor([[0.8] * 5, [0.4] * 5]), lams=0.01, max_iter=5)
    save("g10_pgd", b=b, psf=psf, x_norm1=out, x_nonneg_rhoB=out2)


def g14_other_algorithms():
    """the remaining splitting algorithms on the same kernels (ADMM_vxu, HQS, Pock-Chambolle)."""
    gt, b, psf = synthetic.deconv_case(2, 3, 32, 40, seed=140)
    out = {"b": b, "psf": psf}
    for method in ("admm_vxu", "hqs", "pc"):
        x = dp.Variable()
        fns = dp.sum_squares(dp.conv(x, psf) - T(b)) + dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1)) + dp.nonneg(x)
        out[method] = dp.Problem(fns).solve(method=method, device="cpu", x0=T(b), rhos=0.3, lams=0.01, max_iter=6)
    save("g14_other_algorithms", **out)


def g12_log_descent():
    out = {}
    for tag, kw in (("35_5_30", dict(upper=35, lower=5, iter=30)), ("49_7_24_s", dict(upper=49, lower=7, iter=24, sigma=7.65 / 255)),
                    ("30_10_8_sqrt", dict(upper=30, lower=10, iter=8, sqrt=True, lam=0.1, w=0.7))):
        r, s = log_descent(**kw)
        out[f"rhos_{tag}"], out[f"sigmas_{tag}"] = r, s
    save("g12_log_descent", **out)


def g13_known_answers():
    """The reference's own exact known-answer tests (tests/problem/test_ml_problems.py:5-44), device='cpu'."""
    out = {}
    x = dp.Variable((3, 3))
    rhs = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]])
    dp.Problem(dp.sum_squares(2 * x - rhs)).solve("admm", device="cpu", x0=np.zeros((3, 3)))
    out["lsq"] = x.value
    x = dp.Variable((3, 3))
    dp.Problem(dp.sum_squares(2 * x, rhs)).solve("admm", device="cpu", x0=np.zeros((3, 3)))
    out["lsq1"] = x.value
    x = dp.Variable((3, 3, 1))
    rhs3 = np.array([[[1, 2, 3], [4, 5, 6], [7, 8, 9]]])
    kernel = np.array([[1, 1], [1, 1]]) / 4
    dp.Problem(dp.sum_squares(dp.conv(x, kernel) - rhs3)).solve("admm", device="cpu", x0=np.zeros((3, 3, 1)))
    out["lsq2_x"] = x.value
    out["lsq2_res"] = dp.eval(dp.conv(x, kernel) - rhs3, x.value, zero_out_constant=False)
    x = dp.Variable((3))
    rhs1 = np.array([1, 2, 3])
    dp.Problem(dp.sum_squares(2 * x - rhs1)).solve("admm", device="cpu", x0=np.zeros(3))
    out["lsq3"] = x.value
    save("g13_known_answers", **out)


# --------------------------------------------------------------------------------------
# BASELINE-size cases (configs 2..5 of BASELINE.json): the inputs are regenerated from their seeds by the tests, the
# reference's outputs are stored as strided samples + per-image sums / L2 norms in float64.
# --------------------------------------------------------------------------------------
def _pack(out, key, t, stride):
    t = t.detach()
    out[key] = t[..., ::stride, ::stride].clone()
    d = t.double().reshape(t.shape[0], -1)
    out[key + "_sum"], out[key + "_l2"] = d.sum(1), d.norm(dim=1)


def g30_full_c2():
    """config 2 at its real plane size: 2 of the 8 images (3x1024x1024), ADMM TV-deconv, rho 0.1, lam 0.005, 10 iterations,
    full state at iterations 1 / 5 / 10 (algo/admm.py:49-59, proxfn/sum_square.py:123-156)."""
    gt, b, psf = synthetic.deconv_case(2, 3, 1024, 1024, seed=2302)
    x = dp.Variable()
    fns = _tv_problem(x, T(b), psf)
    out = {"seed": 2302}

    def cb(iter, state, rho, lam):
        if iter + 1 in (1, 5, 10):
            xs, vs, us = state
            _pack(out, f"it{iter + 1}_x", xs, 8)
            for i in range(2):
                _pack(out, f"it{iter + 1}_v{i}", vs[i], 16)
                _pack(out, f"it{iter + 1}_u{i}", us[i], 16)

    xo = dp.Problem(fns).solve(method="admm", device="cpu", x0=T(b), rhos=0.1, lams=0.005, max_iter=10, callback=cb)
    lam10 = np.full(10, 0.005, np.float32)
    x64o, _, _ = admm_f64(b, psf, [("grad0", "norm1", 1.0), ("grad1", "norm1", 1.0)], np.full(10, 0.1, np.float32), [lam10, lam10], 10)
    x64 = ref_f64_admm(b, psf, 10)
    f64_pin(out, "x_f64", x64, x64o)
    _pack(out, "x_f64", x64, 8)
    out["psnr"] = np.array([10 * np.log10(1.0 / np.mean((xo[i].numpy() - gt[i]) ** 2)) for i in range(2)])
    save("g30_full_c2", **out)


def g30b_full_c2_batch8():
    """config 2 exactly as BASELINE.json states it: the whole batch of 8 x 3 x 1024 x 1024, ADMM TV-deconv, rho 0.1, lam 0.005,
    50 iterations -- the final x (::8 samples + per-image sums / L2 norms) and the iterate after 25 iterations, with the float64
    evaluation of the same 50 iterations next to them (algo/admm.py:49-59, proxfn/sum_square.py:123-156)."""
    gt, b, psf = synthetic.deconv_case(8, 3, 1024, 1024, seed=2304)
    x = dp.Variable()
    fns = _tv_problem(x, T(b), psf)
    out = {"seed": 2304}

    def cb(iter, state, rho, lam):
        if iter + 1 == 25:
            _pack(out, "it25_x", state[0], 16)

    xo = dp.Problem(fns).solve(method="admm", device="cpu", x0=T(b), rhos=0.1, lams=0.005, max_iter=50, callback=cb)
    _pack(out, "x", xo, 8)
    out["psnr"] = np.array([10 * np.log10(1.0 / np.mean((xo[i].numpy() - gt[i]) ** 2)) for i in range(8)])
    lam50 = np.full(50, 0.005, np.float32)
    x64o, _, _ = admm_f64(b, psf, [("grad0", "norm1", 1.0), ("grad1", "norm1", 1.0)], np.full(50, 0.1, np.float32), [lam50, lam50], 50)
    x64 = ref_f64_admm(b, psf, 50)
    f64_pin(out, "x_f64", x64, x64o)
    _pack(out, "x_f64", x64, 8)
    out["x_f64"] = out["x_f64"].float()                    # (samples kept in fp32: 6e-8 of the float64 iterate; sums / norms stay float64)
    save("g30b_full_c2_batch8", **out)


def g31_full_c3():
    """config 3 at its real plane size: one 3x1024x1024 image, ADMM with the FFDNet-colour prior (seeded weights), the first
    3 iterations of the log_descent(35, 5, 30) schedule (proxfn/pnp/prior.py:42-89, algo/tune/dpir.py:13-39)."""
    gt, b, psf = synthetic.deconv_case(1, 3, 1024, 1024, seed=2303)
    x = dp.Variable()
    prior = dp.deep_prior(x, denoiser=ColorDen(7))
    fns = dp.sum_squares(dp.conv(x, psf) - T(b)) + prior
    rhos, sigmas = log_descent(35, 5, 30)
    rhos, sigmas = rhos[:3], sigmas[:3]
    out = {"seed": 2303, "rhos": rhos, "sigmas": sigmas}
    with torch.no_grad():
        st = dp.Problem(fns).solve(method="admm", device="cpu", x0=T(b), rhos=rhos, lams={prior: sigmas}, max_iter=3, return_full_states=True)
    _pack(out, "x", st[0], 8)
    _pack(out, "v0", st[1][0], 8)
    _pack(out, "u0", st[2][0], 8)
    x64o, v64o, u64 = admm_f64(b, psf, [("id", "ffdnet", 1.0)], rhos.numpy(), [sigmas.numpy()], 3, ffdnet_weights(7))
    st64 = ref_f64_admm(b, psf, 3, rhos=rhos, lams=sigmas, dims=(), prior=ColorDen(7), full=True)
    f64_pin(out, "x_f64", st64[0], x64o, tol=1e-10)
    f64_pin(out, "v0_f64", st64[1][0], v64o[0], tol=1e-10)
    _pack(out, "x_f64", st64[0], 8)
    _pack(out, "v0_f64", st64[1][0], 8)
    save("g31_full_c3", **out)


def g32_full_c4():
    """config 4, one GPU's shard: 4 x 1 x 320 x 320 CS-MRI, masked-FFT LinOp + nonneg + deep_prior(gray FFDNet), LADMM with the
    CG x-update (rtol 1e-6, <= 100 iterations), 2 outer iterations; the CG exit counts are recorded from the reference's own
    loop (two bdot calls per CG iteration, linalg/solve/solver_cg.py:107-125)."""
    import dprox.linalg.solve.solver_cg as scg
    import dprox.proxfn.sum_square as ssq
    gt, mask, y = synthetic.csmri_case(4, 320, 320, seed=2304, center=32)
    mask, y = T(mask), T(y)
    x = dp.Variable()
    fns = dp.sum_squares(MaskedFFT(x, mask), y) + dp.nonneg(x) + dp.deep_prior(x, denoiser=GrayDen(seed=11))
    x0 = ifft2(y).real.float()
    calls = {"bdot": 0}
    counts = []
    orig_bdot, orig_ls = scg.bdot, ssq.linear_solve

    def bdot(*a, **k):
        calls["bdot"] += 1
        return orig_bdot(*a, **k)

    def linear_solve(*a, **k):
        n0 = calls["bdot"]
        r = orig_ls(*a, **k)
        counts.append((calls["bdot"] - n0) // 2)
        return r

    scg.bdot, ssq.linear_solve = bdot, linear_solve
    try:
        with torch.no_grad():
            st = dp.Problem(fns, linear_solve_config=LinearSolveConfig(rtol=1e-6, max_iters=100)).solve(
                method="ladmm", device="cpu", x0=x0, rhos=0.5, lams=0.03, max_iter=2, return_full_states=True)
    finally:
        scg.bdot, ssq.linear_solve = orig_bdot, orig_ls
    out = {"seed": 2304, "cg_iters": np.array(counts)}
    print("config-4 shard: CG exit counts", counts)
    _pack(out, "x", st[0], 4)
    for i in range(2):
        _pack(out, f"v{i}", st[1][i], 4)
        _pack(out, f"u{i}", st[2][i], 4)
    save("g32_full_c4", **out)


def g32b_full_c4_batches():
    """config 4 at the batch sizes of its 2-GPU and 1-GPU runs: 16 x 1 x 320 x 320 and the full 32 x 1 x 320 x 320 (BASELINE.json
    config 4), same problem and recording as G32 -- 2 outer LADMM iterations, the CG exit counts of the reference's own loop.  These
    batches are beyond the fused CG iteration's size rule (B <= 8), i.e. they pin the step-by-step branch of dpx_cg_masked_fft
    (linalg/solve/solver_cg.py:56-136, proxfn/sum_square.py:158-197)."""
    import dprox.linalg.solve.solver_cg as scg
    import dprox.proxfn.sum_square as ssq
    out = {}
    for B, seed in ((16, 2316), (32, 2332)):
        gt, mask, y = synthetic.csmri_case(B, 320, 320, seed=seed, center=32)
        mask, y = T(mask), T(y)
        x = dp.Variable()
        fns = dp.sum_squares(MaskedFFT(x, mask), y) + dp.nonneg(x) + dp.deep_prior(x, denoiser=GrayDen(seed=11))
        x0 = ifft2(y).real.float()
        calls = {"bdot": 0}
        counts = []
        orig_bdot, orig_ls = scg.bdot, ssq.linear_solve

        def bdot(*a, **k):
            calls["bdot"] += 1
            return orig_bdot(*a, **k)

        def linear_solve(*a, **k):
            n0 = calls["bdot"]
            r = orig_ls(*a, **k)
            counts.append((calls["bdot"] - n0) // 2)
            return r

        scg.bdot, ssq.linear_solve = bdot, linear_solve
        try:
            with torch.no_grad():
                st = dp.Problem(fns, linear_solve_config=LinearSolveConfig(rtol=1e-6, max_iters=100)).solve(
                    method="ladmm", device="cpu", x0=x0, rhos=0.5, lams=0.03, max_iter=2, return_full_states=True)
        finally:
            scg.bdot, ssq.linear_solve = orig_bdot, orig_ls
        print(f"config 4, batch {B}: CG exit counts", counts)
        out[f"B{B}_seed"], out[f"B{B}_cg_iters"] = seed, np.array(counts)
        _pack(out, f"B{B}_x", st[0], 8)
        for i in range(2):
            _pack(out, f"B{B}_v{i}", st[1][i], 8)
            _pack(out, f"B{B}_u{i}", st[2][i], 8)
    save("g32b_full_c4_batches", **out)


def g33_full_c5():
    """config 5 at its real size: 4 x 3 x 512 x 512, ADMM unrolled 10 times (specialize method='unroll'), MSE loss, gradients
    w.r.t. the rho / lambda schedules and the observation through the reference's autograd (specialization/unroll.py:14-58)."""
    gt, b, psf = synthetic.deconv_case(4, 3, 512, 512, seed=2305)
    K = 10
    x = dp.Variable()
    bt = T(b).clone().requires_grad_(True)
    n0, n1 = dp.norm1(dp.grad(x, dim=0)), dp.norm1(dp.grad(x, dim=1))
    solver = dp.compile(dp.sum_squares(dp.conv(x, psf) - bt) + n0 + n1, method="admm", device="cpu")
    solver = dp.specialize(solver, method="unroll", device="cpu", max_iter=K)
    r0, a0, a1 = np.linspace(0.3, 0.1, K).astype("float32"), np.linspace(0.02, 0.005, K).astype("float32"), np.linspace(0.015, 0.006, K).astype("float32")
    rhos, l0, l1 = (torch.tensor(t, requires_grad=True) for t in (r0, a0, a1))
    xo = solver.solve(x0=T(b), rhos=rhos, lams={n0: l0, n1: l1})
    loss = ((xo - T(gt)) ** 2).mean()
    loss.backward()
    out = {"seed": 2305, "rhos": r0, "l0": a0, "l1": a1, "loss": loss.detach().double(), "g_rhos": rhos.grad, "g_l0": l0.grad, "g_l1": l1.grad}
    _pack(out, "x", xo, 8)
    _pack(out, "g_b", bt.grad, 8)
    # the same loss and schedule gradients in float64 (autograd through oracle.admm_f64): the gradients w.r.t. rho_t are sums of
    # 3e6 signed products that cancel to ~1e-5 -- fp32 accumulation order alone moves them by ~1e-4 (relative)
    r64, a64, b64 = (torch.tensor(t, dtype=torch.float64, requires_grad=True) for t in (r0, a0, a1))
    bt64 = T(b).double().requires_grad_(True)
    x64, _, _ = admm_f64(bt64, psf, [("grad0", "norm1", 1.0), ("grad1", "norm1", 1.0)], r64, [a64, b64], K)
    loss64 = ((x64 - T(gt).double()) ** 2).mean()
    loss64.backward()
    # ... and by the REFERENCE itself in float64: its own unrolled solver and autograd (specialization/unroll.py:14-58) on float64
    # tensors inside reference_in_float64(); oracle.admm_f64's figures above must agree with it (f64_pin), the fixture stores the reference's
    with reference_in_float64():
        xr = dp.Variable()
        btr = T64(b).requires_grad_(True)
        m0, m1 = dp.norm1(dp.grad(xr, dim=0)), dp.norm1(dp.grad(xr, dim=1))
        sr = dp.compile(dp.sum_squares(dp.conv(xr, T64(psf)) - btr) + m0 + m1, method="admm", device="cpu")
        sr = dp.specialize(sr, method="unroll", device="cpu", max_iter=K)
        rr, ar, br = (torch.tensor(t, dtype=torch.float64, requires_grad=True) for t in (r0, a0, a1))
        xor = sr.solve(x0=T64(b), rhos=rr, lams={m0: ar, m1: br})
        assert xor.dtype == torch.float64
        lossr = ((xor - T64(gt)) ** 2).mean()
        lossr.backward()
    pins = {}
    for key, ref64, ours in (("x_f64", xor.detach(), x64.detach()), ("g_b_f64", btr.grad, bt64.grad), ("g_rhos_f64", rr.grad, r64.grad),
                             ("g_l0_f64", ar.grad, a64.grad), ("g_l1_f64", br.grad, b64.grad)):
        rel = float((ours - ref64).norm() / ref64.norm())
        print(f"   float64: autograd through oracle.admm_f64 vs the reference's float64 autograd [{key}]: rel-L2 {rel:.2e}")
        assert rel <= 1e-9, (key, rel)          # (the rho gradients are sums of 3e6 signed products that cancel to ~1e-5)
        pins[key + "_oracle_rel"] = np.float64(rel)
    assert abs(float(lossr) - float(loss64)) <= 1e-12 * abs(float(lossr))
    out.update(loss_f64=lossr.detach(), g_rhos_f64=rr.grad, g_l0_f64=ar.grad, g_l1_f64=br.grad, **pins)
    _pack(out, "g_b_f64", btr.grad, 8)
    _pack(out, "x_f64", xor.detach(), 8)         # (pointwise context: after 10 iterations single pixels sit next to threshold decisions)
    print("config 5:", float(loss), rhos.grad, l0.grad, l1.grad)
    print("config 5 f64:", float(loss64), r64.grad, a64.grad, b64.grad)
    save("g33_full_c5", **out)


def g34_pgd_pow2():
    """Proximal gradient descent (algo/pgd.py:26-54) on power-of-two planes, where the backend runs the whole solve as one fused
    call: 2 x 3 x 256 x 512, 8 iterations, decaying per-image step sizes, the three closed-form proximal terms."""
    gt, b, psf = synthetic.deconv_case(2, 3, 256, 512, seed=3401)
    rhos = torch.stack([torch.linspace(0.9, 0.5, 8), torch.linspace(0.6, 0.3, 8)])
    lams = torch.linspace(0.02, 0.005, 8)
    out = {"seed": 3401, "rhos": rhos, "lams": lams}
    for tag, mk in (("norm1", lambda x: 0.7 * dp.norm1(x)), ("nonneg", dp.nonneg), ("norm2", dp.norm2)):
        x = dp.Variable()
        g = mk(x)
        xo = dp.Problem(dp.sum_squares(dp.conv(x, psf) - T(b)) + g).solve(method="pgd", device="cpu", x0=T(b), rhos=rhos, lams={g: lams}, max_iter=8)
        _pack(out, "x_" + tag, xo, 4)
    save("g34_pgd_pow2", **out)


def g35_h768():
    """Column length 768 = 3 * 256 (the height of the reference's own example image, scipy.misc.face 768 x 1024): 1 x 3 x 768 x 256,
    ADMM TV-deconvolution (state after 4 and 10 iterations), the convolution and its adjoint, proximal gradient descent."""
    gt, b, psf = synthetic.deconv_case(1, 3, 768, 256, seed=3501)
    out = {"seed": 3501}
    for K in (4, 10):
        x = dp.Variable()
        st = dp.Problem(_tv_problem(x, T(b), psf)).solve(method="admm", device="cpu", x0=T(b), rhos=0.1, lams=0.005, max_iter=K, return_full_states=True)
        _pack(out, f"it{K}_x", st[0], 8)
        for i in range(2):
            _pack(out, f"it{K}_v{i}", st[1][i], 16)
            _pack(out, f"it{K}_u{i}", st[2][i], 16)
    lam10 = np.full(10, 0.005, np.float32)
    x64o, _, _ = admm_f64(b, psf, [("grad0", "norm1", 1.0), ("grad1", "norm1", 1.0)], np.full(10, 0.1, np.float32), [lam10, lam10], 10)
    x64 = ref_f64_admm(b, psf, 10)
    f64_pin(out, "it10_x_f64", x64, x64o)
    _pack(out, "it10_x_f64", x64, 8)
    x = dp.Variable()
    k2 = gauss(9, 2.5)
    cv = dp.conv(x, k2)
    out["k2"] = k2
    _pack(out, "conv_fwd", cv.forward(T(b)), 8)
    _pack(out, "conv_adj", cv.adjoint(T(b)), 8)
    x = dp.Variable()
    g = dp.norm1(x)
    xo = dp.Problem(dp.sum_squares(dp.conv(x, psf) - T(b)) + g).solve(method="pgd", device="cpu", x0=T(b), rhos=0.8, lams=0.01, max_iter=4)
    _pack(out, "pgd_x", xo, 8)
    save("g35_h768", **out)


def g36_hqs_pow2():
    """Half-quadratic splitting (algo/hqs.py:4-20) on a power-of-two plane, where the backend runs it on the two-kernel ADMM iteration
    with the duals counted as zero: 1 x 3 x 256 x 256, TV + nonneg, 6 iterations, decaying rho; full state (x, v_i)."""
    gt, b, psf = synthetic.deconv_case(1, 3, 256, 256, seed=3601)
    x = dp.Variable()
    fns = dp.sum_squares(dp.conv(x, psf) - T(b)) + dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1)) + dp.nonneg(x)
    rhos = torch.linspace(0.4, 0.2, 6)
    st = dp.Problem(fns).solve(method="hqs", device="cpu", x0=T(b), rhos=rhos, lams=0.01, max_iter=6, return_full_states=True)
    out = {"seed": 3601, "rhos": rhos}
    _pack(out, "x", st[0], 4)
    for i in range(3):
        _pack(out, f"v{i}", st[1][i], 8)
    save("g36_hqs_pow2", **out)


def g37_generic_planes():
    """The plane sizes bench.py times off the register-radix path, at their full size: 8 x 3 x 1000 x 1000 (`other_paths.admm_8x3x1000x1000`:
    the XCD renumbering of the stencil passes and the 8-column interleave only see full grids at this size) and 2 x 3 x 720 x 1280, ADMM
    TV-deconvolution, rho 0.1, lam 0.005, 6 iterations: full state packed like G30 (samples + per-image float64 sums / norms), and the
    float64 iterate of the reference itself (linop/conv.py:31-41 is size-agnostic; algo/admm.py:49-59)."""
    for tag, shape, seed in (("1000", (8, 3, 1000, 1000), 3701), ("720x1280", (2, 3, 720, 1280), 3702)):
        gt, b, psf = synthetic.deconv_case(*shape, seed=seed)
        x = dp.Variable()
        st = dp.Problem(_tv_problem(x, T(b), psf)).solve(method="admm", device="cpu", x0=T(b), rhos=0.1, lams=0.005, max_iter=6, return_full_states=True)
        out = {"seed": seed, "shape": np.array(shape)}
        _pack(out, "x", st[0], 8)
        for i in range(2):
            _pack(out, f"v{i}", st[1][i], 16)
            _pack(out, f"u{i}", st[2][i], 16)
        lam6 = np.full(6, 0.005, np.float32)
        x64o, _, _ = admm_f64(b, psf, [("grad0", "norm1", 1.0), ("grad1", "norm1", 1.0)], np.full(6, 0.1, np.float32), [lam6, lam6], 6)
        x64 = ref_f64_admm(b, psf, 6)
        f64_pin(out, "x_f64", x64, x64o)
        _pack(out, "x_f64", x64, 8)
        out["psnr"] = np.array([10 * np.log10(1.0 / np.mean((st[0][i].numpy() - gt[i]) ** 2)) for i in range(shape[0])])
        save("g37_generic_" + tag, **out)


def g38_full_c3_batch8():
    """config 3 as BASELINE.json states it -- the batch of 8 x 3 x 1024 x 1024, ADMM with the FFDNet-colour prior (seeded weights) on the
    log_descent(35, 5, 30) schedule -- at both ends of the schedule: its first 3 steps from x0 = b (as G31, but the whole batch: the
    launch geometry bench.py times) and its LAST 3 steps (rho, sigma at their smallest: the best-conditioned x-updates and the weakest
    denoising) started from x0 = b as well (27 FFDNet passes over 8 images to get there on the reference's CPU path are out of reach; the
    kernels see the schedule's values, not its history).  x, v, u packed; float64 iterates of the reference itself."""
    gt, b, psf = synthetic.deconv_case(8, 3, 1024, 1024, seed=2308)
    rhos30, sig30 = log_descent(35, 5, 30)
    out = {"seed": 2308}
    for tag, sl in (("first", slice(0, 3)), ("last", slice(27, 30))):
        rhos, sigmas = rhos30[sl].clone(), sig30[sl].clone()
        x = dp.Variable()
        prior = dp.deep_prior(x, denoiser=ColorDen(7))
        fns = dp.sum_squares(dp.conv(x, psf) - T(b)) + prior
        with torch.no_grad():
            st = dp.Problem(fns).solve(method="admm", device="cpu", x0=T(b), rhos=rhos, lams={prior: sigmas}, max_iter=3, return_full_states=True)
        out[tag + "_rhos"], out[tag + "_sigmas"] = rhos, sigmas
        _pack(out, tag + "_x", st[0], 16)
        _pack(out, tag + "_v0", st[1][0], 16)
        _pack(out, tag + "_u0", st[2][0], 16)
        st64 = ref_f64_admm(b, psf, 3, rhos=rhos, lams=sigmas, dims=(), prior=ColorDen(7), full=True)
        _pack(out, tag + "_x_f64", st64[0], 16)
        _pack(out, tag + "_v0_f64", st64[1][0], 16)
        print(f"   g38 {tag}: reference fp32 vs its float64 run: x {float((st[0].double() - st64[0]).norm() / st64[0].norm()):.2e}, "
              f"v {float((st[1][0].double() - st64[1][0]).norm() / st64[1][0].norm()):.2e}")
    save("g38_full_c3_batch8", **out)


def g38b_full_c3_trajectory():
    """config 3 as bench.py times it -- 8 x 3 x 1024 x 1024, ADMM with the FFDNet-colour prior (seeded weights), ALL 30 steps of the
    log_descent(35, 5, 30) schedule in ONE solve from x0 = b -- with x and v sampled at iterations 10, 20 and 30 from the reference and
    from the reference's own float64 run of the same 30 steps (algo/admm.py:49-59, proxfn/pnp/prior.py:42-89, algo/tune/dpir.py:13-39).
    G38 covers the two ends of the schedule from a fresh start; this fixture covers the path in between."""
    gt, b, psf = synthetic.deconv_case(8, 3, 1024, 1024, seed=2308)
    rhos, sigmas = log_descent(35, 5, 30)
    out = {"seed": 2308, "rhos": rhos, "sigmas": sigmas}

    def recorder(suffix):
        def cb(iter, state, rho, lam):
            if iter + 1 in (10, 20, 30):
                _pack(out, f"it{iter + 1}_x{suffix}", state[0], 16)
                _pack(out, f"it{iter + 1}_v0{suffix}", state[1][0], 16)
                _pack(out, f"it{iter + 1}_u0{suffix}", state[2][0], 16)
                print(f"   g38b{suffix}: iteration {iter + 1} recorded", flush=True)
        return cb

    x = dp.Variable()
    prior = dp.deep_prior(x, denoiser=ColorDen(7))
    fns = dp.sum_squares(dp.conv(x, psf) - T(b)) + prior
    with torch.no_grad():
        xo = dp.Problem(fns).solve(method="admm", device="cpu", x0=T(b), rhos=rhos, lams={prior: sigmas}, max_iter=30, callback=recorder(""))
    out["psnr"] = np.array([10 * np.log10(1.0 / np.mean((xo[i].numpy() - gt[i]) ** 2)) for i in range(8)])
    x64 = ref_f64_admm(b, psf, 30, rhos=rhos, lams=sigmas, dims=(), prior=ColorDen(7), callback=recorder("_f64"))
    for it in (10, 20, 30):
        for k in ("x", "v0", "u0"):
            out[f"it{it}_{k}_f64"] = out[f"it{it}_{k}_f64"].float()     # (samples kept in fp32; sums / norms stay float64)
    print(f"   g38b: reference fp32 vs its float64 run after 30 steps: x {float((xo.double() - x64).norm() / x64.norm()):.2e}")
    out["ref_err_x30"] = np.float64(float((xo.double() - x64).norm() / x64.norm()))
    save("g38b_full_c3_trajectory", **out)


def g32c_full_c4_trajectory():
    """config 4, one GPU's shard, at the length bench.py times: 4 x 1 x 320 x 320 CS-MRI, LADMM with the CG x-update (rtol 1e-6, <= 100
    iterations), 10 outer iterations -- the final state and ALL TEN CG exit counts of the reference's own loop
    (linalg/solve/solver_cg.py:99-129, algo/admm.py:78-100); inputs as G32."""
    import dprox.linalg.solve.solver_cg as scg
    import dprox.proxfn.sum_square as ssq
    gt, mask, y = synthetic.csmri_case(4, 320, 320, seed=2304, center=32)
    mask, y = T(mask), T(y)
    x = dp.Variable()
    fns = dp.sum_squares(MaskedFFT(x, mask), y) + dp.nonneg(x) + dp.deep_prior(x, denoiser=GrayDen(seed=11))
    x0 = ifft2(y).real.float()
    calls = {"bdot": 0}
    counts = []
    orig_bdot, orig_ls = scg.bdot, ssq.linear_solve

    def bdot(*a, **k):
        calls["bdot"] += 1
        return orig_bdot(*a, **k)

    def linear_solve(*a, **k):
        n0 = calls["bdot"]
        r = orig_ls(*a, **k)
        counts.append((calls["bdot"] - n0) // 2)
        return r

    out = {"seed": 2304}

    def cb(iter, state, rho, lam):
        if iter + 1 == 5:
            _pack(out, "it5_x", state[0], 4)

    scg.bdot, ssq.linear_solve = bdot, linear_solve
    try:
        with torch.no_grad():
            st = dp.Problem(fns, linear_solve_config=LinearSolveConfig(rtol=1e-6, max_iters=100)).solve(
                method="ladmm", device="cpu", x0=x0, rhos=0.5, lams=0.03, max_iter=10, return_full_states=True, callback=cb)
    finally:
        scg.bdot, ssq.linear_solve = orig_bdot, orig_ls
    out["cg_iters"] = np.array(counts)
    print("config-4 shard, 10 outer iterations: CG exit counts", counts)
    _pack(out, "x", st[0], 4)
    for i in range(2):
        _pack(out, f"v{i}", st[1][i], 4)
        _pack(out, f"u{i}", st[2][i], 4)
    # The yardstick: the reference itself in float64 (reference_in_float64) running the SAME CG iteration counts -- every linear_solve call gets
    # max_iters = the fp32 run's exit count and rtol = 0, so that the two runs differ by float32 round-off alone.  Ten outer iterations of a
    # truncated-CG x-update and a seeded (non-contractive) denoiser amplify that round-off: the reference's own float32 iterate is `ref_err` away
    # from this float64 one (1e-4 class), which is what a second float32 implementation has to be measured against.
    fixed = list(counts)
    k = {"i": 0}

    def linear_solve_fixed(A, b, config=None, *a, **kw):
        cfg = LinearSolveConfig(rtol=0.0, max_iters=int(fixed[k["i"]]))
        k["i"] += 1
        return orig_ls(A, b, cfg, *a, **kw)

    def cb64(iter, state, rho, lam):
        if iter + 1 == 5:
            _pack(out, "it5_x_f64", state[0], 4)

    ssq.linear_solve = linear_solve_fixed
    try:
        with reference_in_float64(), torch.no_grad():
            x64 = dp.Variable()
            y64 = y.to(torch.complex128)
            fns64 = dp.sum_squares(MaskedFFT(x64, mask.double()), y64) + dp.nonneg(x64) + dp.deep_prior(x64, denoiser=GrayDen(seed=11).double())
            st64 = dp.Problem(fns64, linear_solve_config=LinearSolveConfig(rtol=1e-6, max_iters=100)).solve(
                method="ladmm", device="cpu", x0=ifft2(y64).real, rhos=0.5, lams=0.03, max_iter=10, return_full_states=True, callback=cb64)
    finally:
        ssq.linear_solve = orig_ls
    assert st64[0].dtype == torch.float64 and k["i"] == 10
    _pack(out, "x_f64", st64[0], 4)
    for i in range(2):
        _pack(out, f"v{i}_f64", st64[1][i], 4)
        _pack(out, f"u{i}_f64", st64[2][i], 4)
    for kk in list(out):
        if kk.endswith("_f64") and out[kk].dtype == torch.float64:
            out[kk] = out[kk].float()                      # (samples kept in fp32; sums / norms stay float64)
    print(f"   g32c: reference fp32 vs its float64 run (same CG counts) after 10 outer iterations: x {float((st[0].double() - st64[0]).norm() / st64[0].norm()):.2e}, "
          f"after 5: {float((out['it5_x'].double() - out['it5_x_f64'].double()).norm() / out['it5_x_f64'].double().norm()):.2e}")
    save("g32c_full_c4_trajectory", **out)


def g39_train_unrolled_pnp():
    """The training workload bench.py times (`train_unrolled_pnp`; the reference's one published throughput figure, notebooks/quickstart.ipynb:
    254-257) at ITS size: 2 RGB patches of 768 x 768, ADMM unrolled 10 times on sum_squares(conv_doe(x, PSF), b) + deep_prior(ffdnet_color, frozen
    seeded weights), log_descent(49, 7.65, 10) schedules, MSE loss -- forward and the reference's autograd w.r.t. the PSF (through the solver's
    data term AND through the simulated observation b = conv_doe(gt, PSF) + noise), rho_t and sigma_t (algo/specialization/unroll.py:21-58,
    linop/conv.py:59-156).  Inputs as tools/bench_train.py builds them (seed 2024)."""
    from dprox.linop.conv import conv_doe
    bs, size, iters, k = 2, 768, 10, 15
    rng = np.random.RandomState(2024)
    gt = synthetic.synth_detail(rng, bs, 3, size, size)
    psf0 = synthetic.point_spread_function(k, 5.0)
    full = np.zeros((1, 3, size, size), np.float32)
    full[:, :, :k, :k] = psf0[:, :, 0]
    full = np.roll(full, (size // 2 - k // 2, size // 2 - k // 2), axis=(-2, -1))
    noise = (rng.randn(bs, 3, size, size) * 7.65 / 255).astype(np.float32)
    rhos0, sig0 = log_descent(49, 7.65, iters, sigma=7.65 / 255)
    psf = T(full).clone()
    psf = psf / psf.sum(dim=(-2, -1), keepdim=True)
    rhos = rhos0.float().clone().requires_grad_(True)
    lams = sig0.float().clone().requires_grad_(True)            # sigma_t^2; the solver gets the square root
    xv, P, Bv = dp.Variable(), dp.Placeholder(), dp.Placeholder()
    reg = dp.deep_prior(xv, denoiser=ColorDen(7))
    op = conv_doe(xv, P, circular=True)
    solver = dp.compile(dp.sum_squares(op, Bv) + reg, method="admm", device="cpu")
    solver = dp.specialize(solver, method="unroll", device="cpu", max_iter=iters)
    blur = conv_doe(dp.Variable(), P, circular=True)
    P.value = psf
    inp = blur.forward(T(gt)) + T(noise)
    Bv.value = inp
    pred = solver.solve(x0=inp.detach(), rhos=rhos, lams={reg: lams.sqrt()})
    loss = ((pred - T(gt)) ** 2).mean()
    loss.backward()
    g_psf = op.psf.grad + blur.psf.grad          # (the reference wraps the Placeholder's value in one nn.Parameter per operator)
    out = {"seed": 2024, "loss": loss.detach().double(), "g_rhos": rhos.grad, "g_lams": lams.grad, "rhos": rhos0.float(), "lams": sig0.float()}
    _pack(out, "pred", pred.detach(), 8)
    _pack(out, "inp", inp.detach(), 8)
    out["g_psf"] = g_psf[..., size // 2 - 16:size // 2 + 16, size // 2 - 16:size // 2 + 16].clone()      # the 32 x 32 window around the PSF's support
    out["g_psf_l2"] = g_psf.double().norm()
    out["g_psf_sum"] = g_psf.double().sum(dim=(-2, -1))
    print("g39:", float(loss), rhos.grad, lams.grad, float(g_psf.norm()))
    save("g39_train_unrolled_pnp", **out)


if __name__ == "__main__":
    only = sys.argv[1:]
    for fn in (g1_linops, g2_psf2otf, g3_prox, g4_solve_direct, g5_admm_tv, g6_cg, g6b_cg_large_batches, g7_ladmm_cg, g8_ffdnet, g8b_ffdnet_wide_range,
               g9_admm_pnp, g10_pgd, g11_unrolled_grads, g12_log_descent, g13_known_answers, g14_other_algorithms, g15_csmri, g16_ffdnet_grads, g17_mosaic_jd, g18_sisr, g19_conv_doe, g20_drunet, g21_x8_augment, g22_unet, g23_pnp_scaled_sqrt, g24_linear_solve_grad, g25_doe_psf_grad,
               g30_full_c2, g30b_full_c2_batch8, g31_full_c3, g32_full_c4, g32b_full_c4_batches, g33_full_c5, g34_pgd_pow2, g35_h768, g36_hqs_pow2,
               g37_generic_planes, g38_full_c3_batch8, g38b_full_c3_trajectory, g32c_full_c4_trajectory,
               g39_train_unrolled_pnp):
        if not only or any(fn.__name__.startswith(o) for o in only):
            fn()
