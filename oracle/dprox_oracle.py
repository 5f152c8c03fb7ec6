"""TEST INFRASTRUCTURE ONLY -- not part of the product, never imported by ``delta-prox_amd``.

CPU restatement (PyTorch-CPU + NumPy, same op order and the same dtype promotions as
the reference) of the Delta-Prox ADMM / LADMM / PGD iteration hot path, SURVEY.md
section 8(a) rows a1..a13.  Every function cites the reference ``file:line`` it follows
(paths relative to the reference checkout).  It is deliberately written in the
reference's *schedule* (18 full c2c FFTs per TV-deconv ADMM iteration, complex128 divide,
offsets re-evaluated on every prox call) so that

  * it reproduces the reference's fp32 rounding behaviour and can be pinned tightly, and
  * timed on the GPU box's host cores it is a faithful stand-in for the reference's
    PyTorch-CPU path (``bench.py`` cpu_baseline, kind="port").

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real reference in the
build container and stores its outputs under ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function here against those vectors.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

from synthetic import drunet_weights, ffdnet_weights, ircnn_weights, unet_weights  # seeded stand-ins for the un-downloadable checkpoints (shared with tools/)
import torch.nn.functional as F

__all__ = [
    "to_nchw", "psf2otf", "otf_nchw", "grad_kernel", "conv_forward", "conv_adjoint", "conv_diag",
    "Lin", "lin_identity", "lin_conv", "lin_grad", "lin_scale", "lin_custom",
    "Term", "sum_squares", "norm1", "norm2", "nonneg", "deep_prior",
    "soft_threshold", "prox", "LeastSquares", "cg", "bdot", "LinearSolveConfig",
    "solve", "partition_admm", "log_descent", "fft2c", "ifft2c",
    "ffdnet_weights", "ffdnet_forward", "FFDNetOracle", "pixel_unshuffle2", "psnr", "admm_f64",
    "csmri_prox", "custom_admm_csmri", "bayer_mask", "lin_mosaic", "sisr_prox", "admm_ext_prior", "doe_otf", "lin_conv_doe", "drunet_weights", "drunet_forward", "DRUNetOracle", "ircnn_weights", "ircnn_forward", "IRCNNOracle", "AugmentOracle",
    "unet_weights", "unet_forward", "UNetOracle",
]


# --------------------------------------------------------------------------- #
# host-side tensor conversion                                                 #
# --------------------------------------------------------------------------- #
def to_nchw(x, batch=True):
    """``to_torch_tensor`` -- dprox/utils/misc.py:62-96.

    numpy/list -> torch; with ``batch`` a 3-D array whose last dim is 1 or 3 is HWC and is
    permuted to CHW, then a leading batch dim is added while ndim < 4."""
    if isinstance(x, torch.Tensor):
        out = x
    elif isinstance(x, np.ndarray):
        out = torch.tensor(x.copy())
    else:
        out = torch.tensor(x)
    if batch:
        if out.ndim == 3 and out.shape[2] in (1, 3):
            out = out.permute(2, 0, 1)
        if out.ndim < 4:
            out = out.unsqueeze(0)
    return out


# --------------------------------------------------------------------------- #
# a6 / a7: circular convolution by FFT, psf2otf, grad                          #
# --------------------------------------------------------------------------- #
def psf2otf(psf: np.ndarray, outsize: Sequence[int]) -> np.ndarray:
    """MATLAB-style psf2otf -- dprox/utils/psf2otf.py:11-40 (+ helpers :43-98).

    zero-pad 'post' to ``outsize``, circularly shift the PSF centre (floor(size/2)) to the
    origin, n-D FFT over *all* axes (H, W and C), drop the imaginary part if it is within
    ``n_ops`` machine epsilons (np.real_if_close)."""
    psf = np.asarray(psf)
    outsize = np.asarray(outsize)
    while psf.ndim < len(outsize):                       # psf2otf.py:49-50
        psf = psf[..., None]
    psfsize = np.asarray(psf.shape)
    if np.any(psfsize > outsize):                        # psf2otf.py:53-54
        raise ValueError("outsize cannot be smaller than the PSF in any dimension")
    if np.all(psf == 0):
        return np.zeros(tuple(outsize))
    padded = np.zeros(tuple(outsize), dtype=psf.dtype)
    padded[tuple(slice(0, s) for s in psfsize)] = psf    # 'post' padding, psf2otf.py:23-24
    for ax, s in enumerate(psfsize):                     # psf2otf.py:28
        padded = np.roll(padded, -int(s // 2), axis=ax)
    otf = np.fft.fftn(padded)                            # psf2otf.py:29
    n_ops = np.sum(padded.size * np.log2(padded.shape))  # psf2otf.py:34
    return np.real_if_close(otf, tol=n_ops)              # psf2otf.py:35


def _kernel_ndarray(kernel):
    """``to_ndarray`` as used by conv.__init__ -- dprox/utils/misc.py:141-155 / linop/conv.py:19."""
    if isinstance(kernel, torch.Tensor):
        return kernel.detach().cpu().numpy()
    if isinstance(kernel, np.ndarray):
        return kernel.astype("float32")
    return np.array(kernel)


def otf_nchw(kernel, shape) -> torch.Tensor:
    """``conv._FB`` -- dprox/linop/conv.py:23-29 (psf2otf over [H, W, C], then batchify)."""
    _, C, H, W = shape
    FB = torch.from_numpy(psf2otf(_kernel_ndarray(kernel), [H, W, C]))
    if FB.ndim == 3 and FB.shape[2] in (1, 3):           # batchify, utils/misc.py:56-57
        FB = FB.permute(2, 0, 1)
    return FB.unsqueeze(0)


def grad_kernel(dim: int) -> torch.Tensor:
    """``grad.__init__`` -- dprox/linop/grad.py:14-21: int64 [1,-1] along H(0)/W(1)/C(2) of an HWC kernel."""
    if dim not in (0, 1, 2):
        raise ValueError("dim must be 0(Height) or 1(Width) or 2 (Channel)")
    D = torch.tensor([1, -1]).unsqueeze(0).unsqueeze(0)
    return D.transpose(dim, -1)


def conv_forward(x: torch.Tensor, FB: torch.Tensor) -> torch.Tensor:
    """``conv.forward`` -- dprox/linop/conv.py:31-35."""
    Fx = torch.fft.fftn(x, dim=[-2, -1])
    return torch.real(torch.fft.ifftn(FB * Fx, dim=[-2, -1])).float()


def conv_adjoint(x: torch.Tensor, FB: torch.Tensor) -> torch.Tensor:
    """``conv.adjoint`` -- dprox/linop/conv.py:37-41."""
    Fx = torch.fft.fftn(x, dim=[-2, -1])
    return torch.real(torch.fft.ifftn(torch.conj(FB) * Fx, dim=[-2, -1])).float()


def conv_diag(FB: torch.Tensor) -> torch.Tensor:
    """``conv.get_diag`` -- dprox/linop/conv.py:46-53: |OTF|^2 (float32 or float64 like FB)."""
    return torch.abs(torch.conj(FB) * FB)


# --------------------------------------------------------------------------- #
# a8: the linear-operator DAG, flattened to "one linop applied to x, plus a    #
# constant" (everything the hot path uses: conv(x)-b, grad(x), x, s*x - b,     #
# user-defined forward/adjoint pairs)                                          #
# --------------------------------------------------------------------------- #
@dataclass
class Lin:
    """value(x) = fwd(x) + const     (LinOp.value, dprox/linop/base.py:109-115)

    ``diag(shape_ref, freq)`` mirrors ``get_diag``; ``gram_diag_space`` / ``gram_diag_freq`` mirror
    ``is_gram_diag(freq=False/True)`` (linop/base.py:56-62, conv.py:43-44, variable.py:40-44,
    sum.py:36-39, scale.py:38-46, blackbox.py:69-72)."""
    fwd: Callable[[torch.Tensor], torch.Tensor]
    adj: Callable[[torch.Tensor], torch.Tensor]
    diag: Optional[Callable] = None
    gram_diag_space: bool = False
    gram_diag_freq: bool = False
    const: Optional[torch.Tensor] = None

    def minus(self, b) -> "Lin":
        """``linop - b`` = sum([linop, Constant(-b)]) -- linop/base.py:196-208,232-234 (b is NOT batchified)."""
        c = -(b if isinstance(b, torch.Tensor) else torch.tensor(b))
        return Lin(self.fwd, self.adj, self.diag, self.gram_diag_space, self.gram_diag_freq, c)

    def value(self, x):
        """sum.forward -- linop/sum.py:13-19: zeros_like(first) then += each input."""
        y = self.fwd(x)
        if self.const is None:
            return y
        out = torch.zeros_like(y)
        out += y
        out += self.const
        return out

    def dag_forward(self, x):
        """CompGraph(linop, zero_out_constant=True).forward -- comp_graph.py:54-57,198-220."""
        y = self.fwd(x)
        if self.const is None:
            return y
        out = torch.zeros_like(y)
        out += y
        out += self.const * 0
        return out

    def offset(self, xref):
        """LinOp.offset -- linop/base.py:117-129: value with every Variable zeroed (re-runs the DAG)."""
        return self.value(torch.zeros_like(xref))


def lin_identity() -> Lin:
    """Variable -- linop/variable.py:22-59 (get_diag = ones)."""
    return Lin(lambda x: x, lambda y: y, lambda ref, freq: torch.ones(ref.shape), True, True)


def lin_conv(kernel) -> Lin:
    """conv(x, kernel) -- linop/conv.py:15-56 (OTF cached per input shape)."""
    cache = {}

    def FB(shape):
        shape = tuple(shape)
        if shape not in cache:
            cache[shape] = otf_nchw(kernel, shape)
        return cache[shape]

    return Lin(lambda x: conv_forward(x, FB(x.shape)), lambda y: conv_adjoint(y, FB(y.shape)),
               lambda ref, freq: conv_diag(FB(ref.shape)), False, True)


def lin_grad(dim=1) -> Lin:
    """grad(x, dim) -- linop/grad.py:8-23 (a conv with the int64 [1,-1] kernel => complex128 OTF)."""
    return lin_conv(grad_kernel(dim))


def lin_scale(s: float, inner: Lin) -> Lin:
    """scale -- linop/scale.py:20-61 (get_diag = (s*d)*conj(s*d))."""
    def diag(ref, freq):
        d = inner.diag(ref, freq) * s
        return d * torch.conj(d)
    return Lin(lambda x: inner.fwd(x) * s, lambda y: inner.adj(y * s), diag,
               inner.gram_diag_space, inner.gram_diag_freq)


def doe_otf(psf: torch.Tensor, shape) -> torch.Tensor:
    """psf2otf2 -- linop/conv.py:59-78 (padding split from the height difference, ifftshift over all dims, fft2)."""
    _, _, fh, fw = psf.shape
    if shape[2] != fh:
        pad = (shape[2] - fh) / 2
        if (shape[2] - fh) % 2 != 0:
            pt = pl = int(np.ceil(pad)); pb = pr = int(np.floor(pad))
        else:
            pt = pl = int(pad) + 1; pb = pr = int(pad) - 1
        psf = torch.nn.functional.pad(psf, [pl, pr, pt, pb], mode="constant")
    return torch.fft.fft2(torch.fft.ifftshift(psf))


def lin_conv_doe(psf, circular=True) -> "Lin":
    """conv_doe -- linop/conv.py:81-148: forward real(ifftn(otf * fftn(x))), adjoint with conj(otf), diag |otf|^2.
    ``circular=False`` (conv.py:100-108,121-129): zero-pad to 2H x 2H (both axes from the height), same product, crop."""
    psf = torch.as_tensor(psf).float()

    def apply(x, conj):
        crop = None
        if not circular:
            side = 2 * x.shape[2]
            hp, wp = (side - x.shape[2]) / 2, (side - x.shape[3]) / 2
            pt, pb, pl, pr = int(np.ceil(hp)), int(np.floor(hp)), int(np.ceil(wp)), int(np.floor(wp))
            x = torch.nn.functional.pad(x, [pl, pr, pt, pb], mode="constant")
            crop = (pt, pb, pl, pr)
        otf = doe_otf(psf, x.shape)
        out = torch.real(torch.fft.ifftn((torch.conj(otf) if conj else otf) * torch.fft.fftn(x, dim=[-2, -1]), dim=[-2, -1])).float()
        if crop is not None:
            out = out[:, :, crop[0]:-crop[1], crop[2]:-crop[3]]
        return out

    def diag(x, freq):
        o = doe_otf(psf, x.shape)
        return torch.abs(torch.conj(o) * o)
    return Lin(lambda x: apply(x, False), lambda x: apply(x, True), diag, False, True)        # is_diag(freq=True) only: conv.py:136-137


def bayer_mask(H, W) -> torch.Tensor:
    """linop/subsample.py:33-46: RGGB colour-filter-array mask as [1,3,H,W] float32."""
    m = np.zeros((3, H, W), dtype=np.float32)
    for ch, (y, x) in zip((0, 1, 1, 2), [(0, 0), (0, 1), (1, 0), (1, 1)]):
        m[ch, y::2, x::2] = 1
    return torch.from_numpy(m)[None]


def lin_mosaic(inner: "Lin") -> "Lin":
    """mosaic(inner) -- subsample.py:17-31: y = mask * inner(x), adjoint = inner^T(mask * y); not diagonalisable once
    composed with a convolution, so the x-update goes through CG."""
    def fwd(x):
        y = inner.fwd(x)
        return bayer_mask(*y.shape[-2:]) * y

    def adj(y):
        return inner.adj(bayer_mask(*y.shape[-2:]) * y)
    return lin_custom(fwd, adj)


def lin_custom(forward, adjoint, diag=None) -> Lin:
    """A user LinOp subclass / LinOpFactory black box -- linop/blackbox.py:4-72, linop/base.py:56-62."""
    return Lin(forward, adjoint, (lambda ref, freq: diag(ref)) if diag is not None else None,
               diag is not None, diag is not None)


# --------------------------------------------------------------------------- #
# a9 / a10: proximal operators                                                 #
# --------------------------------------------------------------------------- #
@dataclass(eq=False)
class Term:
    """One ``ProxFn(linop)`` of the objective -- proxfn/base.py:30-41 (alpha from ``scalar * fn``, :78-82)."""
    kind: str
    lin: Lin
    alpha: float = 1.0
    beta: float = 1.0
    denoise: Optional[Callable] = None       # deep_prior only: (x[B,C,H,W], sigma) -> x
    sqrt: bool = False
    clamp: bool = False
    b: Optional[object] = None               # sum_squares(linop, b) form -- proxfn/sum_square.py:15-24

    def offset(self, xref):
        """ProxFn.offset / sum_squares.offset -- proxfn/base.py:43-45, sum_square.py:20-24."""
        if self.b is not None:
            return to_nchw(self.b, batch=True)
        return -self.lin.offset(xref)


def sum_squares(lin, b=None, alpha=1.0):
    return Term("sum_squares", lin, alpha, b=b)


def norm1(lin, alpha=1.0):
    return Term("norm1", lin, alpha)


def norm2(lin, alpha=1.0):
    return Term("norm2", lin, alpha)


def nonneg(lin):
    return Term("nonneg", lin)


def deep_prior(lin, denoise, sqrt=False, clamp=False):
    return Term("deep_prior", lin, denoise=denoise, sqrt=sqrt, clamp=clamp)


def soft_threshold(v, lam):
    """proxfn/norm.py:6-11."""
    return torch.sign(v) * torch.maximum(torch.abs(v) - lam, torch.zeros_like(v))


def _prox_raw(term: Term, v, lam):
    if term.kind == "norm1":                              # proxfn/norm.py:18-19
        return soft_threshold(v, lam)
    if term.kind in ("norm2", "sum_squares"):             # proxfn/norm.py:26-27, sum_square.py:26-27
        return v / (1 + 2 * lam)
    if term.kind == "nonneg":                             # proxfn/nonneg.py:10-11
        return torch.maximum(v, torch.zeros_like(v))
    if term.kind == "deep_prior":                         # proxfn/pnp/prior.py:73-86
        sigma = torch.sqrt(torch.clamp(lam, min=1e-8)) if term.sqrt else lam   # utils/misc.py:158-161
        if term.clamp:
            v = v.clamp(0, 1)
        if torch.is_complex(v):
            v = v.real
        inp = v.unsqueeze(1) if v.ndim == 3 else v
        out = term.denoise(inp, sigma.view(-1, 1, 1, 1))  # Denoiser.denoise, denoisers/base.py:6-10
        return out.type_as(v).reshape(*v.shape)
    raise ValueError(term.kind)


def prox(term: Term, v, lam, xref):
    """``ProxFn.prox`` -- proxfn/base.py:55-64 with prox_scaled/affine/translated (:12-27).

    (1/beta) * P(beta*(v - off), beta^2 * lam * alpha) + off,  off = -linop.offset re-evaluated per call."""
    if lam.ndim == 1:
        lam = lam.view(lam.shape[0], 1, 1, 1)
    off = term.offset(xref)                               # proxfn/base.py:43-45
    a, b = term.alpha, term.beta
    return 1.0 / b * _prox_raw(term, b * (v - off), (b * b * lam) * a) + off


# --------------------------------------------------------------------------- #
# a5: matrix-free CG                                                           #
# --------------------------------------------------------------------------- #
@dataclass
class LinearSolveConfig:
    """linalg/custom.py:9-26."""
    rtol: float = 1e-6
    max_iters: int = 100
    verbose: bool = False
    solver_type: str = "cg"
    solver_kwargs: dict = field(default_factory=dict)
    use_analytic_grad: bool = True


def bdot(x, y):
    """linalg/solve/solver_cg.py:7-22."""
    if x.ndim == 1:
        return torch.dot(x, y)
    return torch.sum(x.reshape(x.shape[0], -1) * y.reshape(y.shape[0], -1), dim=-1)


def _expand(g, ref):
    while g.ndim < ref.ndim:                              # solver_cg.py:25-39
        g = g.unsqueeze(-1)
    return g


def _ravel(x):
    return x if x.ndim == 1 else x.reshape(x.shape[0], -1)   # solver_cg.py:42-53


def cg(A, b, x0=None, rtol=1e-6, max_iters=100, return_iters=False):
    """linalg/solve/solver_cg.py:56-136.  NB ``torch.linalg.norm(ravel(r), 2)`` on a [B,N]
    matrix is the spectral norm (:103), so the stop rule couples the images of a batch."""
    x = torch.zeros_like(b) if x0 is None else x0
    r = A(x)
    r *= -1.0
    r += b
    cg_tol = rtol * torch.linalg.norm(_ravel(b), 2, dim=-1)          # :95
    gamma_1 = p = None
    n_it = int(np.minimum(max_iters, np.prod(b.shape)))              # :99
    done = n_it
    for it in range(n_it):
        normr = torch.linalg.norm(_ravel(r), 2)                      # :103
        if torch.all(normr <= cg_tol):                               # :104
            done = it
            break
        gamma = _expand(bdot(r, r), x)
        if it > 0:
            beta = gamma / gamma_1
            p = r + beta * p
        else:
            p = r
        q = A(p)
        alpha = gamma / _expand(bdot(p, q), x)
        x = x + alpha * p
        r = r - alpha * q
        gamma_1 = gamma
    return (x, done) if return_iters else x


# --------------------------------------------------------------------------- #
# a4: the x-update                                                             #
# --------------------------------------------------------------------------- #
class LeastSquares:
    """``least_squares`` -- proxfn/sum_square.py:87-197 (chosen by algo/invert.py:5-14)."""

    def __init__(self, quad: List[Term], other: List[Term], try_diagonalize=True,
                 try_freq_diagonalize=True, linear_solve_config: LinearSolveConfig = None):
        self.quad, self.other = quad, other
        lins = [t.lin for t in quad + other]
        self.diagonalizable = all(l.gram_diag_space for l in lins) and try_diagonalize          # :106
        self.freq_diagonalizable = (all(l.gram_diag_freq for l in lins)
                                    and try_diagonalize and try_freq_diagonalize)               # :107
        self.cfg = linear_solve_config or LinearSolveConfig()
        self.cg_iters = []

    def solve(self, b, rho, xref, v=None, eps=1e-7):
        if rho.ndim == 1:                                                                       # :116-117
            rho = rho.view(rho.shape[0], 1, 1, 1)
        if self.diagonalizable or self.freq_diagonalizable:
            return self.solve_direct(b, rho, xref, v, eps)
        return self.solve_cg(b, rho, xref, v)

    def _ktb(self, b, rho, xref, v):
        Ktb = 0
        for t in self.quad:                               # :126-132: adjoint of the *offset*, re-evaluated
            Ktb += t.lin.adj(t.offset(xref))
        for i, t in enumerate(self.other):
            Ktb += rho * t.lin.adj(b[i])
        if v is not None:
            Ktb += rho * v
        return Ktb

    def solve_direct(self, b, rho, xref, v=None, eps=1e-7):
        """proxfn/sum_square.py:123-156."""
        Ktb = self._ktb(b, rho, xref, v)
        freq = self.freq_diagonalizable
        diag = 0
        for t in self.quad:
            diag = diag + t.lin.diag(Ktb, freq)
        for t in self.other:
            diag = diag + rho * t.lin.diag(Ktb, freq)
        if v is not None:
            diag = diag + rho
        if freq:
            FK = torch.fft.fftn(Ktb, dim=[-2, -1])
            out = torch.real(torch.fft.ifftn((FK + eps) / (diag + eps), dim=[-2, -1]))
        else:
            out = Ktb / (diag + eps)
        return out.float()

    def solve_cg(self, b, rho, xref, v=None):
        """proxfn/sum_square.py:158-197 -> linalg.linear_solve -> cg."""
        def KtK(x):
            out = 0
            for t in self.quad:
                out += t.lin.adj(t.lin.dag_forward(x))
            for t in self.other:
                out += rho * t.lin.adj(t.lin.dag_forward(x))
            if v is not None:
                out += rho * x
            return out
        Ktb = self._ktb(b, rho, xref, v)
        x, n = cg(KtK, Ktb, rtol=self.cfg.rtol, max_iters=self.cfg.max_iters, return_iters=True)
        self.cg_iters.append(n)
        return x


# --------------------------------------------------------------------------- #
# a1 / a2 / a3 / a11: the iteration drivers                                    #
# --------------------------------------------------------------------------- #
def partition_admm(terms: List[Term]):
    """ADMM.partition -- algo/admm.py:26-36 (exact-type sum_squares -> Omega, the rest -> Psi)."""
    omega = [t for t in terms if t.kind == "sum_squares"]
    psi = [t for t in terms if t.kind != "sum_squares"]
    return psi, omega


def _isscalar(x):
    return np.isscalar(x) or (isinstance(x, torch.Tensor) and x.ndim == 0)   # algo/base.py:54-55


def _defaults(psi, rhos, lams, max_iter):
    """Algorithm.defaults -- algo/base.py:205-218 (rho=1.0, lam=0.02; scalar lam -> Psi terms only)."""
    if rhos is None:
        rhos = 1.0
    if lams is None:
        lams = 0.02
    if _isscalar(rhos):
        rhos = torch.tensor([float(rhos)] * max_iter)
    if _isscalar(lams):
        lams = {t: torch.tensor([float(lams)] * max_iter) for t in psi}
    lams = {k: (torch.tensor([float(v)] * max_iter) if _isscalar(v) else torch.as_tensor(v))
            for k, v in lams.items()}
    return torch.as_tensor(rhos), lams


def solve(terms: List[Term], method="admm", x0=None, rhos=None, lams=None, max_iter=24,
          callback=None, return_full_states=False, try_diagonalize=True, try_freq_diagonalize=True,
          linear_solve_config: LinearSolveConfig = None, return_solver=False):
    """``Problem.solve`` -> ``compile`` -> ``Algorithm.solve`` -> ``iters``
    (algo/problem.py:46-55, algo/primitives.py:40-67, algo/base.py:85-156)."""
    x0 = to_nchw(x0, batch=True)
    if method == "pgd":
        return _solve_pgd(terms, x0, rhos, lams, max_iter, callback, return_full_states)
    psi, omega = partition_admm(terms)
    rhos, lams = _defaults(psi, rhos, lams, max_iter)
    ls = LeastSquares(omega, psi, try_diagonalize, try_freq_diagonalize, linear_solve_config)

    # ADMM.initialize -- algo/admm.py:61-67
    x = x0
    v = [t.lin.value(x) for t in psi]              # K = CompGraph(vstack(...)) keeps constants (algo/base.py:78)
    u = [torch.zeros_like(e) for e in v]
    state = (x, v, u)
    for it in range(max_iter):                       # algo/base.py:149-156
        rho = rhos[..., it]
        lam = {k: val[..., it] for k, val in lams.items()}
        x, v, u = state
        if method == "admm":                         # ADMM._iter -- algo/admm.py:49-59
            b = [v[i] - u[i] for i in range(len(psi))]
            x = ls.solve(b, rho, xref=x)
            Kx = [t.lin.value(x) for t in psi]
        elif method == "ladmm":                      # LinearizedADMM._iter -- algo/admm.py:78-100
            b = []
            for i, t in enumerate(psi):
                tmp = t.lin.dag_forward(x) - v[i] + u[i]
                b.append(x - 1 * t.lin.adj(tmp))
            x = ls.solve(b, rho, xref=x)
            Kx = [t.lin.value(x) for t in psi]
            if len(psi) == 1:                        # algo/admm.py:95 -- no return_list: Kx[i] indexes the batch dim
                Kx = Kx[0]
        else:
            raise ValueError(method)
        for i, t in enumerate(psi):
            v[i] = prox(t, Kx[i] + u[i], lam[t], xref=x)
            u[i] = u[i] + Kx[i] - v[i]
        state = (x, v, u)
        if callback is not None:
            callback(iter=it, state=state, rho=rho, lam=lam)
    if return_solver:
        return state, ls
    return state if return_full_states else state[0]


def _solve_pgd(terms, x0, rhos, lams, max_iter, callback, return_full_states):
    """ProximalGradientDescent -- algo/pgd.py:8-54; sum_squares.grad -- proxfn/sum_square.py:29-32."""
    if len(terms) != 2:
        raise ValueError("Proximal gradient descent only supports two proximal functions for now.")
    omega = [t for t in terms if t.kind == "sum_squares"]      # hasattr(fn, 'grad'), pgd.py:15-17
    psi = [t for t in terms if t not in omega]
    if not omega:
        raise ValueError("Proximal gradient descent requires at least one proximal function is differentiable.")
    diff, pfn = omega[0], psi[0]
    rhos, lams = _defaults(psi, rhos, lams, max_iter)
    x = x0
    for it in range(max_iter):
        rho = rhos[..., it]
        lam = {k: val[..., it] for k, val in lams.items()}
        r = rho.view(rho.shape[0], 1, 1, 1) if rho.ndim == 1 else rho
        g = diff.lin.adj(diff.lin.dag_forward(x) - diff.offset(x))
        vv = x - r * g
        x = prox(pfn, vv, lam[pfn], xref=x)
        if callback is not None:
            callback(iter=it, state=[x], rho=rho, lam=lam)
    return [x] if return_full_states else x


# --------------------------------------------------------------------------- #
# a12 / a13: schedules and the centred orthonormal FFT                         #
# --------------------------------------------------------------------------- #
def log_descent(upper, lower, iter=24, sigma=0.255 / 255, w=1.0, lam=0.23, sqrt=False):
    """algo/tune/dpir.py:13-39."""
    s_log = np.logspace(np.log10(upper), np.log10(lower), iter).astype(np.float32)
    s_lin = np.linspace(upper, lower, iter).astype(np.float32)
    sigmas = (s_log * w + s_lin * (1 - w)) / 255.0
    rhos = [lam * (sigma ** 2) / (s ** 2) for s in sigmas]
    if not sqrt:
        sigmas = list(sigmas ** 2)
    return torch.tensor(rhos).float(), torch.tensor(np.asarray(sigmas)).float()


def fft2c(x):
    """utils/misc.py:164-177."""
    x = torch.fft.ifftshift(x, dim=(-2, -1))
    x = torch.fft.fft2(x, norm="ortho")
    return torch.fft.fftshift(x, dim=(-2, -1))


def ifft2c(x):
    """utils/misc.py:180-193."""
    x = torch.fft.ifftshift(x, dim=(-2, -1))
    x = torch.fft.ifft2(x, norm="ortho")
    return torch.fft.fftshift(x, dim=(-2, -1))


# --------------------------------------------------------------------------- #
# a10: FFDNet                                                                  #
# --------------------------------------------------------------------------- #
def pixel_unshuffle2(x):
    """models/basicblock.py:104-126 with upscale_factor 2: out channel = c*4 + dy*2 + dx."""
    B, C, H, W = x.shape
    v = x.contiguous().view(B, C, H // 2, 2, W // 2, 2)
    return v.permute(0, 1, 3, 5, 2, 4).contiguous().view(B, C * 4, H // 2, W // 2)


def ffdnet_forward(x, sigma, layers):
    """``FFDNet.forward`` -- models/network_ffdnet.py:54-68 (conv stack built by basicblock.py:61-98)."""
    h, w = x.shape[-2:]
    pad_b = int(np.ceil(h / 2) * 2 - h)
    pad_r = int(np.ceil(w / 2) * 2 - w)
    x = F.pad(x, (0, pad_r, 0, pad_b), mode="replicate")
    x = pixel_unshuffle2(x)
    m = torch.ones((x.shape[0], 1, x.shape[2], x.shape[3])) * sigma.view(-1, 1, 1, 1)
    x = torch.cat((x, m), 1)
    for i, (wt, bs) in enumerate(layers):
        x = F.conv2d(x, torch.as_tensor(wt), torch.as_tensor(bs), padding=1)
        if i < len(layers) - 1:
            x = F.relu(x)
    x = F.pixel_shuffle(x, 2)
    return x[..., :h, :w]


class FFDNetOracle:
    """``FFDNetColorDenoiser`` (denoisers/wrapper.py:38-48) or, with ``per_band=True``, the gray
    ``FFDNetDenoiser`` applied band-by-band by ``Denoiser2D`` (wrapper.py:25-35, denoisers/base.py:17-25)."""

    def __init__(self, layers, per_band=False):
        self.layers, self.per_band = layers, per_band

    def __call__(self, x, sigma):
        sigma = sigma.view(-1, 1, 1, 1)
        if not self.per_band:
            return ffdnet_forward(x, sigma, self.layers)
        return torch.cat([ffdnet_forward(band, sigma, self.layers) for band in x.split(1, dim=1)], dim=1)


class AugmentOracle:
    """``Augment`` -- denoisers/composite.py:6-46 (``deep_prior(x8=True)``, prior.py:55-56): call number k applies the
    dihedral transform k % 8 before the denoiser and its inverse afterwards (3 and 5 invert each other, :18-21)."""

    def __init__(self, base):
        self.base, self.iter = base, 0

    @staticmethod
    def augment(img, mode):
        # composite.py:31-46, mode by mode
        if mode == 0:
            return img
        if mode == 1:
            return torch.flip(torch.rot90(img, 1, [2, 3]), [2])
        if mode == 2:
            return torch.flip(img, [2])
        if mode == 3:
            return torch.rot90(img, 3, [2, 3])
        if mode == 4:
            return torch.flip(torch.rot90(img, 2, [2, 3]), [2])
        if mode == 5:
            return torch.rot90(img, 1, [2, 3])
        if mode == 6:
            return torch.rot90(img, 2, [2, 3])
        return torch.flip(torch.rot90(img, 3, [2, 3]), [2])

    def __call__(self, x, sigma):
        m = self.iter % 8
        y = self.base(self.augment(x, m), sigma)
        self.iter += 1
        return self.augment(y, 8 - m if m in (3, 5) else m)


# --------------------------------------------------------------------------- #
# CS-MRI pipeline: closed-form data term + CustomADMM                         #
# --------------------------------------------------------------------------- #
def csmri_prox(v, lam, num_psi, mask, y):
    """proxfn/fast/csmri.py:14-25: z = fft2(v); z[mask] = ((lam z + y)/(1 + lam num_psi))[mask]; ifft2(z)."""
    lam = torch.as_tensor(lam, dtype=torch.float32)
    if lam.ndim == 1:
        lam = lam.view(lam.shape[0], 1, 1, 1)
    mask = mask.bool()
    z = fft2c(v)
    temp = ((lam * z.clone()) + y) / (1 + lam * num_psi)
    z[mask] = temp[mask]
    return ifft2c(z)


def custom_admm_csmri(x0, y, mask, rhos, sigmas, max_iter, denoise):
    """contrib/csmri.py:156-171 (CustomADMM._iter) driven by algo/base.py:128-156 with one deep_prior Psi term and the
    csmri data term routed through ext_sum_squares.solve (sum_square.py:35-48, invert.py:8-12); state initialised by
    admm.py:61-67 (z = [x0], u = [0]).  deep_prior takes the real part of a complex iterate (pnp/prior.py:79).
    Returns (x, z, u) after max_iter iterations."""
    x, z, u = x0, x0.clone(), torch.zeros_like(x0)
    for i in range(max_iter):
        rho, sigma = rhos[..., i], sigmas[..., i]
        v = z - u
        if torch.is_complex(v):
            v = v.real
        x = denoise(v.contiguous(), torch.as_tensor(sigma, dtype=torch.float32).reshape(-1)).type_as(v)
        b = x + u
        z = csmri_prox(b, rho, 1, mask, y)
        u = u + x - z
    return x, z, u


# --------------------------------------------------------------------------- #
# closed-form super-resolution data term                                      #
# --------------------------------------------------------------------------- #
def _sr_splits(a, sf):
    """proxfn/fast/sr.py:83-93."""
    b = torch.stack(torch.chunk(a, sf, dim=2), dim=4)
    return torch.cat(torch.chunk(b, sf, dim=3), dim=4)


def sisr_prox(v, lam, I, y, kernel, sf):
    """proxfn/fast/sr.py:52-77 (reload + _prox): kernel [1,1,kh,kw] tensor, y [N,C,h,w]."""
    h, w = y.shape[-2:]
    STy = torch.zeros(y.shape[0], y.shape[1], h * sf, w * sf, dtype=y.dtype)              # upsample, sr.py:117-126
    STy[..., 0::sf, 0::sf] = y
    otf = torch.zeros(kernel.shape[:-2] + (h * sf, w * sf), dtype=kernel.dtype)           # p2o, sr.py:95-114
    otf[..., :kernel.shape[2], :kernel.shape[3]] = kernel
    for axis, axis_size in enumerate(kernel.shape[2:]):
        otf = torch.roll(otf, -int(axis_size / 2), dims=axis + 2)
    FB = torch.fft.fftn(otf, dim=(-2, -1))
    FBC, F2B = torch.conj(FB), torch.pow(torch.abs(FB), 2)
    FBFy = FBC * torch.fft.fftn(STy, dim=(-2, -1))
    FR = FBFy + torch.fft.fftn(lam * v, dim=(-2, -1))
    x1 = FB.mul(FR)
    FBR = torch.mean(_sr_splits(x1, sf), dim=-1, keepdim=False)
    invW = torch.mean(_sr_splits(F2B, sf), dim=-1, keepdim=False)
    invWBR = FBR.div(invW + I * lam)
    FCBinvWBR = FBC * invWBR.repeat(1, 1, sf, sf)
    FX = (FR - FCBinvWBR) / (I * lam + 1e-9)
    return torch.real(torch.fft.ifftn(FX, dim=(-2, -1)))


def admm_ext_prior(x0, ext_prox, denoise, rhos, sigmas, max_iter):
    """ADMM (algo/admm.py:49-67) for  ext_sum_squares data term + one deep prior on x: the x-update is the data term's
    own closed form (sum_square.py:44-48 via invert.py:8-12).  Returns (x, v, u)."""
    x, v, u = x0, x0.clone(), torch.zeros_like(x0)
    for i in range(max_iter):
        rho, sigma = rhos[..., i], sigmas[..., i]
        x = ext_prox(v - u, rho, 1)
        d = x + u
        v = denoise(d.contiguous(), torch.as_tensor(sigma, dtype=torch.float32).reshape(-1))
        u = d - v
    return x, v, u


# --------------------------------------------------------------------------- #
# DRUNet (UNetRes) denoiser                                                    #
# --------------------------------------------------------------------------- #
def drunet_forward(x0, sd, nb=4):
    """UNetRes.forward -- models/network_unet.py:106-117 with ResBlock = x + conv(relu(conv(x))) (basicblock.py:211-223),
    2x2 stride-2 convolutions down (:437-443) and 2x2 stride-2 transposed convolutions up (:413-419)."""
    import torch.nn.functional as F

    def res(x, pre):
        for i in range(nb):
            i0 = i if not pre.startswith("m_up") else i + 1
            x = x + F.conv2d(F.relu(F.conv2d(x, sd[f"{pre}.{i0}.res.0.weight"], padding=1)), sd[f"{pre}.{i0}.res.2.weight"], padding=1)
        return x
    x1 = F.conv2d(x0, sd["m_head.weight"], padding=1)
    x2 = F.conv2d(res(x1, "m_down1"), sd[f"m_down1.{nb}.weight"], stride=2)
    x3 = F.conv2d(res(x2, "m_down2"), sd[f"m_down2.{nb}.weight"], stride=2)
    x4 = F.conv2d(res(x3, "m_down3"), sd[f"m_down3.{nb}.weight"], stride=2)
    x = res(x4, "m_body")
    x = res(F.conv_transpose2d(x + x4, sd["m_up3.0.weight"], stride=2), "m_up3")
    x = res(F.conv_transpose2d(x + x3, sd["m_up2.0.weight"], stride=2), "m_up2")
    x = res(F.conv_transpose2d(x + x2, sd["m_up1.0.weight"], stride=2), "m_up1")
    return F.conv2d(x + x1, sd["m_tail.weight"], padding=1)


class DRUNetOracle:
    """DRUNetDenoiser -- denoisers/wrapper.py:89-146: sigma map as an extra channel, replicate-pad to a multiple of 16 for
    images up to 256x256, otherwise four overlapping quadrants (recursively), each denoised on its own."""

    def __init__(self, sd):
        self.sd = sd

    def __call__(self, x, sigma):
        sigma = sigma.view(-1, 1, 1, 1)
        if sigma.shape[0] != x.shape[0]:
            sigma = sigma.repeat(x.shape[0], 1, 1, 1)
        L = torch.cat((x, sigma.repeat(1, 1, x.shape[2], x.shape[3])), dim=1)
        return self._denoise(L)

    def _denoise(self, L, refield=32, min_size=256, modulo=16):
        h, w = L.shape[-2:]
        if h * w <= min_size ** 2:
            Lp = torch.nn.functional.pad(L, (0, int(np.ceil(w / modulo) * modulo - w), 0, int(np.ceil(h / modulo) * modulo - h)), mode="replicate")
            return drunet_forward(Lp, self.sd)[..., :h, :w]
        top, bottom = slice(0, (h // 2 // refield + 1) * refield), slice(h - (h // 2 // refield + 1) * refield, h)
        left, right = slice(0, (w // 2 // refield + 1) * refield), slice(w - (w // 2 // refield + 1) * refield, w)
        Ls = [L[..., top, left], L[..., top, right], L[..., bottom, left], L[..., bottom, right]]
        if h * w <= 4 * (min_size ** 2):
            Es = [drunet_forward(q, self.sd) for q in Ls]
        else:
            Es = [self._denoise(q, refield, min_size, modulo) for q in Ls]
        b, c = Es[0].shape[:2]
        E = torch.zeros(b, c, h, w, dtype=L.dtype)
        E[..., :h // 2, :w // 2] = Es[0][..., :h // 2, :w // 2]
        E[..., :h // 2, w // 2:] = Es[1][..., :h // 2, (-w + w // 2):]
        E[..., h // 2:, :w // 2] = Es[2][..., (-h + h // 2):, :w // 2]
        E[..., h // 2:, w // 2:] = Es[3][..., (-h + h // 2):, (-w + w // 2):]
        return E


def ircnn_forward(x, sd):
    """IRCNN.forward -- models/network_dncnn.py:94-113: x - model(x), seven 3x3 convolutions with dilations 1,2,3,4,3,2,1
    (padding = dilation), ReLU between them."""
    import torch.nn.functional as F
    n = x
    for i, d in enumerate((1, 2, 3, 4, 3, 2, 1)):
        n = F.conv2d(n, sd[f"model.{2 * i}.weight"], sd[f"model.{2 * i}.bias"], padding=d, dilation=d)
        if i < 6:
            n = F.relu(n)
    return x - n


class IRCNNOracle:
    """IRCNNDenoiser -- denoisers/wrapper.py:68-86 (Denoiser2D: band by band): one of 25 models chosen by
    ceil(sigma * 255 / 2) - 1."""

    def __init__(self, model25):
        self.model25 = model25

    def __call__(self, x, sigma):
        idx = int(np.ceil(sigma.reshape(-1)[:1].cpu().numpy() * 255. / 2.)[0] - 1)      # float32 arithmetic, like the reference
        sd = self.model25[str(idx)]
        return torch.cat([ircnn_forward(band, sd) for band in x.split(1, dim=1)], dim=1)


def psnr(out, gt):
    """utils/metrics.py:68-70 restated for data range 1: 10*log10(1/MSE), per image."""
    mse = ((out - gt) ** 2).reshape(out.shape[0], -1).mean(dim=1)
    return 10.0 * torch.log10(1.0 / mse)


# --------------------------------------------------------------------------- #
# float64 evaluation of the same ADMM iteration ("exact iterate")             #
# --------------------------------------------------------------------------- #
def admm_f64(b, psf, psi, rhos, lams, max_iter, ffdnet_layers=None):
    """The ADMM iteration of algo/admm.py:49-59 for  sum_squares(conv(x, psf) - b) + sum_i g_i(K_i x)  evaluated in
    float64 with exact OTFs: the iterate both the reference's fp32 path and the HIP path approximate.

    Used to put parity numbers in context: when rho * |G|^2 + |H|^2 gets tiny (PnP schedules reach 1e-5) the x-update
    amplifies fp32 round-off by 1/min(den), and the reference's own output is then 1e-4..1e-3 away from this iterate.

    psi: list of (linop, prox, alpha) with linop in {"id", "grad0", "grad1"}, prox in {"norm1", "nonneg", "ffdnet"};
    rhos: [T]; lams: list (one per psi term) of [T] arrays; x0 = b.  Returns (x, [v_i], [u_i]) float64 tensors."""
    bb = torch.as_tensor(b).double()
    B, C, H, W = bb.shape

    def otf(k):
        return torch.from_numpy(np.transpose(psf2otf(np.asarray(k, dtype=np.float64), [H, W, C]), (2, 0, 1))[None])

    Hf = otf(_kernel_ndarray(psf).astype(np.float64))
    G = {"grad0": otf(grad_kernel(0).numpy()), "grad1": otf(grad_kernel(1).numpy())}
    F2 = lambda a: torch.fft.fftn(a, dim=[-2, -1])
    Fi = lambda a: torch.real(torch.fft.ifftn(a, dim=[-2, -1]))
    K = lambda name, a: a if name == "id" else Fi(G[name] * F2(a))
    Kt = lambda name, a: a if name == "id" else Fi(torch.conj(G[name]) * F2(a))
    gram = lambda name: torch.ones_like(Hf.real) if name == "id" else torch.abs(G[name]) ** 2

    def den64(x, sigma):
        l64 = [(torch.as_tensor(w).double(), torch.as_tensor(bs).double()) for w, bs in ffdnet_layers]
        h, w = x.shape[-2:]
        xx = F.pad(x, (0, w % 2, 0, h % 2), mode="replicate")
        xx = pixel_unshuffle2(xx)
        m = torch.ones((xx.shape[0], 1, xx.shape[2], xx.shape[3]), dtype=torch.float64) * sigma.view(-1, 1, 1, 1)
        xx = torch.cat((xx, m), 1)
        for i, (wt, bs) in enumerate(l64):
            xx = F.conv2d(xx, wt, bs, padding=1)
            if i < len(l64) - 1:
                xx = F.relu(xx)
        return F.pixel_shuffle(xx, 2)[..., :h, :w]

    x = bb.detach().clone()                  # x0 = b as a separate tensor (the reference's solve(x0=...) does not tie it to b)
    v = [K(name, x) for name, _, _ in psi]
    u = [torch.zeros_like(e) for e in v]
    Ktb = Fi(torch.conj(Hf) * F2(bb))
    for it in range(max_iter):
        # (tensor schedules stay tensors so that torch.autograd can differentiate the float64 iteration w.r.t. them)
        rho = rhos[it].double() if isinstance(rhos, torch.Tensor) else float(np.float32(rhos[it]))
        rhs = Ktb + rho * sum(Kt(name, v[i] - u[i]) for i, (name, _, _) in enumerate(psi))
        den = torch.abs(Hf) ** 2 + rho * sum(gram(name) for name, _, _ in psi)
        x = Fi((F2(rhs) + 1e-7) / (den + 1e-7))
        for i, (name, prox_kind, alpha) in enumerate(psi):
            lam = (lams[i][it].double() if isinstance(lams[i], torch.Tensor) else float(np.float32(lams[i][it]))) * alpha
            d = K(name, x) + u[i]
            if prox_kind == "norm1":
                v[i] = torch.sign(d) * torch.clamp(d.abs() - lam, min=0)
            elif prox_kind == "nonneg":
                v[i] = torch.clamp(d, min=0)
            else:
                v[i] = den64(d, torch.full((B,), float(lam), dtype=torch.float64))
            u[i] = d - v[i]
    return x, v, u


# --------------------------------------------------------------------------- #
# U-Net denoiser                                          #
# --------------------------------------------------------------------------- #
def unet_forward(x, sd):
    """``UNet.forward`` -- dprox/proxfn/pnp/denoisers/models/unet/unet.py:47-64 with its blocks: ConvBlock = 3 x (Conv2d 3x3
    pad 1 + bias + LeakyReLU(0.2)) (:8-31), down = MaxPool2d(2) + ConvBlock (:76-84), up = bilinear x2 (align_corners=True),
    zero-pad to the skip tensor's size (left/top = diff // 2), cat([skip, up]), ConvBlock (:89-117), outc = 1x1 conv (:121-128);
    output = input[:, :C_out] + residual (:61-64).  ``sd``: the reference state dict."""
    def block(t, prefix):
        for i in range(3):
            t = F.leaky_relu(F.conv2d(t, sd[f"{prefix}.conv-{i}.conv2d.weight"], sd[f"{prefix}.conv-{i}.conv2d.bias"], padding=1), 0.2)
        return t

    def up(lo, skip, prefix):
        lo = F.interpolate(lo, scale_factor=2, mode="bilinear", align_corners=True)
        dy, dx = skip.shape[2] - lo.shape[2], skip.shape[3] - lo.shape[3]
        lo = F.pad(lo, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))
        return block(torch.cat([skip, lo], dim=1), prefix)

    x1 = block(x, "inc.conv")
    x2 = block(F.max_pool2d(x1, 2), "down1.mpconv.1")
    x3 = block(F.max_pool2d(x2, 2), "down2.mpconv.1")
    x4 = block(F.max_pool2d(x3, 2), "down3.mpconv.1")
    x5 = block(F.max_pool2d(x4, 2), "down4.mpconv.1")
    t = up(x5, x4, "up1.conv")
    t = up(t, x3, "up2.conv")
    t = up(t, x2, "up3.conv")
    t = up(t, x1, "up4.conv")
    res = F.conv2d(t, sd["outc.conv.weight"], sd["outc.conv.bias"])
    return x[:, :res.shape[1]] + res


class UNetOracle:
    """``UNetDenoiser`` -- denoisers/wrapper.py:206-221 (a Denoiser2D: band by band, noise map as second channel, clamp to [0, 1])"""

    def __init__(self, sd):
        self.sd = sd

    def denoise(self, x, sigma):
        sigma = sigma.view(-1, 1, 1, 1)
        outs = []
        for band in x.split(1, dim=1):
            noise_map = torch.ones_like(band) * sigma
            outs.append(torch.clamp(unet_forward(torch.cat([band, noise_map], dim=1), self.sd), 0, 1))
        return torch.cat(outs, dim=1)
