"""Tensor-level wrappers around the C ABI (device pointers + current stream from PyTorch-ROCm).

PyTorch is used for device memory and streams only; every arithmetic pass below is a HIP kernel
of ``libdpx_hip.so``.
"""
import ctypes
from ctypes import c_double, c_float, c_void_p

import numpy as np
import torch

from . import _backend as be
from ._backend import Term, ptr, require

EPS = 1e-7   # least_squares.solve default (reference dprox/proxfn/sum_square.py:115)


# ----------------------------------------------------------------------------------------------
# cached tables / workspaces
# ----------------------------------------------------------------------------------------------
_tables = {}
_workspaces = {}


def _bytes(n, device):
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device=device)


def fft_table(H, W, device):
    key = (str(device), H, W)
    t = _tables.get(key)
    if t is None:
        L = be.lib()
        t = _bytes(L.query("dpx_fft_table_bytes", H, W), device)
        L.call("dpx_fft_table_init", ptr(t), H, W, be.stream())
        _tables[key] = t
    return t


_zero_scalars = {}


def zero_scalar(device):
    """a cached 0-d zero on ``device`` (a stand-in for an absent offset in the differentiable path: never written)"""
    key = str(device)
    if key not in _zero_scalars:
        _zero_scalars[key] = torch.zeros((), device=device)
    return _zero_scalars[key]


def workspace(tag, nbytes, device):
    key = (str(device), tag)
    w = _workspaces.get(key)
    if w is None or w.numel() < nbytes:
        w = _bytes(nbytes, device)
        _workspaces[key] = w
    return w


def spectrum_ws(P, H, W, device):
    return workspace("spectrum", be.lib().query("dpx_spectrum_bytes", P, H, W), device)


def clear_caches():
    _tables.clear()
    _workspaces.clear()
    _psf_dev.clear()
    _dd_cache.clear()


def _shape4(x):
    if x.ndim != 4:
        raise be.DpxError(f"expected an NCHW tensor, got shape {tuple(x.shape)}")
    return tuple(int(s) for s in x.shape)


def as_batch_vec(v, B, device):
    """scalar / 0-d / [B] / [B,1,1,1] -> contiguous float32 [B] on device."""
    if not isinstance(v, torch.Tensor):
        v = torch.tensor(float(v))
    v = v.detach().to(device=device, dtype=torch.float32).reshape(-1)
    if v.numel() == 1:
        v = v.expand(B)
    elif v.numel() != B:
        raise be.DpxError(f"per-image scalar has {v.numel()} entries for a batch of {B}")
    return v.contiguous()


# ----------------------------------------------------------------------------------------------
# OTF tables
# ----------------------------------------------------------------------------------------------
_psf_dev = {}           # (kernel bytes, shape, device) -> fp64 device copy: a kernel is uploaded once, not once per table built from it


def psf_to_device(psf, device):
    """kernel (2-D, or HWC 3-D) -> contiguous fp64 [kh,kw,kc] on device."""
    k = np.asarray(psf, dtype=np.float64)
    if k.ndim == 1:
        k = k[None, :, None]
    elif k.ndim == 2:
        k = k[:, :, None]
    elif k.ndim != 3:
        raise ValueError(f"kernel must be 1-D, 2-D or HWC 3-D, got shape {k.shape}")
    k = np.ascontiguousarray(k)
    if k.nbytes <= 1 << 16:
        key = (k.tobytes(), k.shape, str(device))
        hit = _psf_dev.get(key)
        if hit is None:
            if len(_psf_dev) >= 64:
                _psf_dev.pop(next(iter(_psf_dev)))
            hit = _psf_dev[key] = torch.from_numpy(k).to(device)     # (a pageable upload blocks the host behind everything queued: once)
        return hit, k.shape
    return torch.from_numpy(k).to(device), k.shape


def make_otf(psf, C, H, W, device):
    """complex OTF table for dpx_fft_conv (opaque layout)."""
    L = be.lib()
    kd, (kh, kw, kc) = psf_to_device(psf, device)
    if kh > H or kw > W or kc > C:
        raise ValueError(f"outsize {[H, W, C]} cannot be smaller than the PSF array size {[kh, kw, kc]} in any dimension.")
    otf = _bytes(L.query("dpx_otf_bytes", C, H, W), device)
    L.call("dpx_psf2otf", ptr(kd), kh, kw, kc, C, H, W, ptr(otf), None, c_float(1.0), 0, be.stream())
    return otf


def new_diag(C, H, W, device):
    d = torch.zeros(max(be.lib().query("dpx_diag_bytes", C, H, W) // 4, 4), dtype=torch.float32, device=device)
    return d


def accumulate_diag(diag, psf, weight, C, H, W):
    """diag += weight * |OTF(psf)|^2"""
    L = be.lib()
    kd, (kh, kw, kc) = psf_to_device(psf, diag.device)
    L.call("dpx_psf2otf", ptr(kd), kh, kw, kc, C, H, W, None, ptr(diag), c_float(weight), 1, be.stream())
    return diag


def diag_to_full(diag, C, H, W):
    """opaque diag table -> full [1,C,H,W] |OTF|^2 array on the table's device (setup-time interop)"""
    full = torch.empty(1, C, H, W, dtype=torch.float32, device=diag.device)
    be.lib().call("dpx_table_to_full", ptr(diag), ptr(full), C, H, W, be.stream())
    return full


def otf_from_full(full, C, H, W):
    """complex64 OTF on the full grid [.., C, H, W] -> opaque half-spectrum OTF table for fft_conv"""
    f = full.reshape(-1, C, H, W)[0].to(torch.complex64).contiguous()
    require(f, dtype=torch.complex64, what="full OTF")
    tab = _bytes(be.lib().query("dpx_otf_bytes", C, H, W), f.device)
    be.lib().call("dpx_otf_from_full", ptr(f), ptr(tab), C, H, W, be.stream())
    return tab


def diag_from_full(full, C, H, W, device):
    """full-spectrum real diagonal [.., C, H, W] -> opaque table (user-supplied BlackBox diagonals)"""
    f = torch.as_tensor(full).real.float().reshape(-1, C, H, W)[0].contiguous().to(device)
    tab = new_diag(C, H, W, device)
    be.lib().call("dpx_table_from_full", ptr(f), ptr(tab), C, H, W, be.stream())
    return tab


_dd_cache = {}


def denominator(d0, c0, d1, c1, C, H, W, device):
    """interleaved (d0 + c0, d1 + c1) table for fourier_solve; cached on the identity of its inputs"""
    key = (None if d0 is None else (d0.data_ptr(), d0._version), float(c0), None if d1 is None else (d1.data_ptr(), d1._version),
           float(c1), C, H, W, str(device))
    hit = _dd_cache.get(key)
    if hit is None:
        L = be.lib()
        dd = _bytes(L.query("dpx_denominator_bytes", C, H, W), device)
        L.call("dpx_denominator_pack", ptr(d0), c_float(c0), ptr(d1), c_float(c1), ptr(dd), C, H, W, be.stream())
        if len(_dd_cache) > 64:
            _dd_cache.clear()
        hit = (dd, d0, d1)            # keep the sources alive so data_ptr keys stay unique
        _dd_cache[key] = hit
    return hit[0]


# ----------------------------------------------------------------------------------------------
# Fourier-domain operators
# ----------------------------------------------------------------------------------------------
def fft_conv(x, otf, conj=False, out=None):
    require(x, what="fft_conv input")
    B, C, H, W = _shape4(x)
    y = torch.empty_like(x) if out is None else out
    be.lib().call("dpx_fft_conv", ptr(x), ptr(y), ptr(otf), int(bool(conj)), B, C, H, W,
                  ptr(fft_table(H, W, x.device)), ptr(spectrum_ws(B * C, H, W, x.device)), be.stream())
    return y


def pgd_supported(H, W, kind):
    return bool(be.lib().query("dpx_pgd_supported", int(H), int(W), int(kind)))


def pgd_run(x, ktb, gram_otf, kind, alpha, rho_tab, lam_tab, T, ws=None):
    """T fused proximal-gradient iterations on x, in place (two kernels per iteration; rho_tab / lam_tab: [T, B] device tables).
    ws: a spectrum workspace of the caller's (concurrent calls on different streams must not share the cached one)"""
    require(x, what="pgd iterate")
    B, C, H, W = _shape4(x)
    if ws is None:
        ws = spectrum_ws(B * C, H, W, x.device)
    be.lib().call("dpx_pgd_run", ptr(x), ptr(ktb), ptr(gram_otf), int(kind), c_float(alpha), ptr(rho_tab), ptr(lam_tab), int(T),
                  B, C, H, W, ptr(fft_table(H, W, x.device)), ptr(ws), be.stream())
    return x


def data_spectrum(b, otf=None, conj=True, out=None, accumulate=False):
    """packed fp32 half spectrum of op(OTF) * F(b), transform evaluated in fp64 (once per solve)"""
    require(b, what="data_spectrum input")
    B, C, H, W = _shape4(b)
    L = be.lib()
    if out is None:
        out = _bytes(L.query("dpx_spectrum_bytes", B * C, H, W) // 2, b.device)
        accumulate = False
    ws = workspace("data_spectrum", L.query("dpx_data_spectrum_ws_bytes", B * C, H, W), b.device)
    L.call("dpx_data_spectrum", ptr(b), ptr(otf), int(bool(conj)), ptr(out), int(bool(accumulate)), B, C, H, W, ptr(ws), be.stream())
    return out


def fourier_solve(rhs, d0, d1, c0, c1, rho, eps=EPS, out=None, spec_add=None):
    require(rhs, what="fourier_solve rhs")
    B, C, H, W = _shape4(rhs)
    rho = as_batch_vec(rho, B, rhs.device)
    x = torch.empty_like(rhs) if out is None else out
    dd = denominator(d0, c0, d1, c1, C, H, W, rhs.device)
    be.lib().call("dpx_fourier_solve", ptr(rhs), ptr(x), ptr(spec_add), ptr(dd), ptr(rho),
                  c_float(eps), B, C, H, W, ptr(fft_table(H, W, rhs.device)),
                  ptr(spectrum_ws(B * C, H, W, rhs.device)), be.stream())
    return x


def fourier_apply_inv(g, d0, d1, c0, c1, rho, eps=EPS):
    """irFFT2[rFFT2(g) / (d0 + c0 + rho (d1 + c1) + eps)]: the self-adjoint linear part of fourier_solve (its backward)"""
    require(g, what="fourier_apply_inv input")
    B, C, H, W = _shape4(g)
    rho = as_batch_vec(rho, B, g.device)
    out = torch.empty_like(g)
    dd = denominator(d0, c0, d1, c1, C, H, W, g.device)
    be.lib().call("dpx_fourier_apply_inv", ptr(g), ptr(out), ptr(dd), ptr(rho), c_float(eps), B, C, H, W,
                  ptr(fft_table(H, W, g.device)), ptr(spectrum_ws(B * C, H, W, g.device)), be.stream())
    return out


def prox_bwd(kind, d, g, lam, alpha=1.0, off=None, want_dlam=True):
    """(J(d)^T g, d prox / d lam at d) of ops.prox -- see dpx_prox_bwd"""
    require(d, what="prox_bwd point")
    require(g, what="prox_bwd gradient")
    B = int(d.shape[0])
    npb = d.numel() // max(B, 1)
    lam_v = as_batch_vec(lam, B, d.device)
    gd = torch.empty_like(d)
    dl = torch.empty_like(d) if want_dlam else None
    be.lib().call("dpx_prox_bwd", int(kind), ptr(d), ptr(g), ptr(gd), ptr(dl), ptr(lam_v), c_float(alpha), ptr(off), B, npb, be.stream())
    return gd, dl


# ----------------------------------------------------------------------------------------------
# spatial / elementwise
# ----------------------------------------------------------------------------------------------
def grad(x, dim, adjoint=False):
    require(x, what="grad input")
    B, C, H, W = _shape4(x)
    y = torch.empty_like(x)
    be.lib().call("dpx_grad", ptr(x), ptr(y), int(dim), int(bool(adjoint)), B, C, H, W, be.stream())
    return y


def _flat_real(t):
    return torch.view_as_real(t) if t.is_complex() else t


def lincomb(terms, out=None):
    """out = sum_i coef_i * x_i, terms = [(coef, x)] with coef a python float or a per-image [B] tensor.
    Real fp32 tensors, or complex64 tensors with real coefficients (handled as interleaved floats)."""
    xs = [t[1] for t in terms]
    ref = xs[0]
    if ref.dtype == torch.float64:
        return _lincomb_f64(terms, out)
    cplx = ref.is_complex()
    for x in xs:
        require(x, dtype=torch.complex64 if cplx else torch.float32, what="lincomb operand")
        if x.shape != ref.shape:
            raise be.DpxError(f"lincomb: shape mismatch {tuple(x.shape)} vs {tuple(ref.shape)}")
    res = torch.empty_like(ref) if out is None else out
    B = int(ref.shape[0]) if ref.ndim > 0 else 1
    npb = (ref.numel() // max(B, 1)) * (2 if cplx else 1)
    if ref.numel() == 0:
        return res
    i, first = 0, True
    while i < len(terms):
        take = 4 if first else 3
        ops = ([] if first else [(1.0, res)]) + list(terms[i:i + take])
        i += take
        first = False
        n = len(ops)
        px = (c_void_p * n)(*[_flat_real(o[1]).data_ptr() for o in ops])
        cf = (c_float * n)()
        pb = (c_void_p * n)()
        keep = []
        for j, (c, _) in enumerate(ops):
            if isinstance(c, torch.Tensor) and c.numel() > 1:
                cb = as_batch_vec(c, B, ref.device)
                keep.append(cb)
                cf[j] = 1.0
                pb[j] = cb.data_ptr()
            else:
                cf[j] = float(c)
                pb[j] = None
        be.lib().call("dpx_lincomb", ptr(_flat_real(res)), n, px, cf, pb, B, npb, be.stream())
    return res


def _lincomb_f64(terms, out=None):
    """float64 operands (the Krylov solvers keep a float64 system in float64): dpx_lincomb_f64"""
    ref = terms[0][1]
    for _, x in terms:
        require(x, dtype=torch.float64, what="lincomb operand")
        if x.shape != ref.shape:
            raise be.DpxError(f"lincomb: shape mismatch {tuple(x.shape)} vs {tuple(ref.shape)}")
    res = torch.empty_like(ref) if out is None else out
    if ref.numel() == 0:
        return res
    B = int(ref.shape[0]) if ref.ndim > 0 else 1
    npb = ref.numel() // max(B, 1)
    i, first = 0, True
    while i < len(terms):
        take = 4 if first else 3
        grp = ([] if first else [(1.0, res)]) + list(terms[i:i + take])
        i += take
        first = False
        n = len(grp)
        px = (c_void_p * n)(*[o[1].data_ptr() for o in grp])
        cf, pb, keep = (c_double * n)(), (c_void_p * n)(), []
        for j, (c, _) in enumerate(grp):
            if isinstance(c, torch.Tensor) and c.numel() > 1:
                cb = c.detach().to(device=ref.device, dtype=torch.float64).reshape(-1).contiguous()
                if cb.numel() != B:
                    raise be.DpxError(f"per-image scalar has {cb.numel()} entries for a batch of {B}")
                keep.append(cb)
                cf[j], pb[j] = 1.0, cb.data_ptr()
            else:
                cf[j], pb[j] = float(c), None
        be.lib().call("dpx_lincomb_f64", ptr(res), n, px, cf, pb, B, npb, be.stream())
    return res


def absmax(x):
    """max |x| over all elements (float32 or float64) as a 0-d tensor on x's device -- pcg's stop rule"""
    f64 = x.dtype == torch.float64
    require(x, dtype=torch.float64 if f64 else torch.float32, what="absmax operand")
    out = torch.empty((), dtype=x.dtype, device=x.device)
    ws = workspace("absmax", 256 * 8, x.device)
    be.lib().call("dpx_absmax", ptr(x), ptr(out), x.numel(), 1 if f64 else 0, ptr(ws), be.stream())
    return out


def bdot(x, y):
    if x.dtype == torch.float64:
        require(x, dtype=torch.float64, what="bdot x"), require(y, dtype=torch.float64, what="bdot y")
        B = int(x.shape[0])
        npb = x.numel() // B
        L = be.lib()
        out = torch.empty(B, dtype=torch.float64, device=x.device)
        ws = workspace("dot64", L.query("dpx_bdot_f64_ws_bytes", B, npb), x.device)
        L.call("dpx_bdot_f64", ptr(x), ptr(y), ptr(out), B, npb, ptr(ws), be.stream())
        return out
    require(x, what="bdot x"), require(y, what="bdot y")
    B = int(x.shape[0])
    npb = x.numel() // B
    L = be.lib()
    out = torch.empty(B, dtype=torch.float32, device=x.device)
    ws = workspace("dot", L.query("dpx_bdot_ws_bytes", B, npb), x.device)
    L.call("dpx_bdot", ptr(x), ptr(y), ptr(out), B, npb, ptr(ws), be.stream())
    return out


def bgram(r):
    if r.dtype == torch.float64:
        require(r, dtype=torch.float64, what="bgram r")
        B = int(r.shape[0])
        npb = r.numel() // B
        L = be.lib()
        out = torch.empty(B, B, dtype=torch.float64, device=r.device)
        ws = workspace("dot64", L.query("dpx_bdot_f64_ws_bytes", B, npb), r.device)
        L.call("dpx_bgram_f64", ptr(r), ptr(out), B, npb, ptr(ws), be.stream())
        return out
    require(r, what="bgram r")
    B = int(r.shape[0])
    npb = r.numel() // B
    L = be.lib()
    out = torch.empty(B, B, dtype=torch.float32, device=r.device)
    ws = workspace("dot", L.query("dpx_bdot_ws_bytes", B, npb), r.device)
    L.call("dpx_bgram", ptr(r), ptr(out), B, npb, ptr(ws), be.stream())
    return out


def cg_masked_fft(b, mask, rho, n_identity, rtol, max_iters):
    """the whole CG x-update of a masked-Fourier data term in one C call (dpx_cg_masked_fft); returns (x, exit iteration)"""
    require(b, what="cg right-hand side")
    B = int(b.shape[0])
    H, W = int(b.shape[-2]), int(b.shape[-1])
    assert b.numel() == B * H * W, "one plane per system"
    mask = mask.to(device=b.device, dtype=torch.float32).contiguous()
    mimg = B if mask.numel() == b.numel() else 1
    assert mask.numel() == mimg * H * W
    L = be.lib()
    x = torch.empty_like(b)
    ws = workspace("cg_masked_fft", L.query("dpx_cg_masked_fft_ws_bytes", B, H, W, mimg), b.device)
    n = L.query("dpx_cg_masked_fft", ptr(x), ptr(b), ptr(mask), mimg, ptr(as_batch_vec(rho, B, b.device)), c_float(n_identity), c_float(rtol),
                int(max_iters), B, H, W, ptr(fft_table(H, W, b.device)), ptr(ws), be.stream())
    if n < 0:
        raise be.DpxError(f"dpx_cg_masked_fft failed ({n}): {L.cdll.dpx_last_error().decode()}")
    return x, int(n)


def zeros_like(t):
    """torch.zeros_like through the C ABI's stream memset"""
    out = torch.empty_like(t)
    if be.host_mode():
        return out.zero_()
    be.lib().call("dpx_zero", ptr(out), out.numel() * out.element_size(), be.stream())
    return out


class CgControl:
    """device-resident control block of one cg() solve (dpx_cg_*): stop rule, beta / alpha and the iterate updates run on the
    GPU; the host only issues kernels and polls the `done` flag without blocking"""

    MAX_B = 64

    def __init__(self, b, rtol):
        L = be.lib()
        self.B = int(b.shape[0])
        self.npb = b.numel() // self.B
        self.dev = b.device
        self.state = torch.empty(L.query("dpx_cg_state_bytes", self.B) // 4, dtype=torch.float32, device=b.device)
        self.flags = self.state[5 * self.B:].view(torch.int32)                  # done, n_done, it, pad
        self.pAp = self.state[3 * self.B:4 * self.B]
        self.gram = torch.empty(self.B, self.B, dtype=torch.float32, device=b.device)
        self.ws = workspace("dot", L.query("dpx_bdot_ws_bytes", self.B, self.npb), b.device)
        self.bnorm2 = bdot(b, b)                                                # <b_i, b_i> (kept alive: the call below is asynchronous)
        L.call("dpx_cg_init", ptr(self.state), ptr(self.bnorm2), c_float(rtol), self.B, be.stream())

    def test(self, r):
        L = be.lib()
        L.call("dpx_bgram", ptr(r), ptr(self.gram), self.B, self.npb, ptr(self.ws), be.stream())
        L.call("dpx_cg_test", ptr(self.state), ptr(self.gram), self.B, be.stream())

    def direction(self, p, r):
        be.lib().call("dpx_cg_direction", ptr(p), ptr(r), ptr(self.state), self.B, self.npb, be.stream())

    def update(self, x, r, p, Ap):
        L = be.lib()
        L.call("dpx_bdot", ptr(p), ptr(Ap), ptr(self.pAp), self.B, self.npb, ptr(self.ws), be.stream())
        L.call("dpx_cg_update", ptr(x), ptr(r), ptr(p), ptr(Ap), ptr(self.state), self.B, self.npb, be.stream())


def prox(kind, v, lam, alpha=1.0, off=None, out=None):
    require(v, what="prox input")
    B = int(v.shape[0])
    npb = v.numel() // B
    lam_v = None if lam is None else as_batch_vec(lam, B, v.device)
    if off is not None:
        off = require(off.expand_as(v).contiguous(), what="prox offset")
    res = torch.empty_like(v) if out is None else out
    be.lib().call("dpx_prox", int(kind), ptr(v), ptr(res), ptr(lam_v), c_float(alpha), ptr(off), B, npb, be.stream())
    return res


def _bwd_ws(B, C, H, W, device):
    return workspace("admm_bwd", be.lib().query("dpx_admm_bwd_ws_bytes", B, C, H, W), device)


def admm_zupdate_bwd(specs, shape, device):
    """specs: dict(linop, prox, alpha, lam[B], v, gv|None, gu_new|None) per term -> (gx, [gu_i], [glam_i [B]])"""
    B, C, H, W = shape
    n = len(specs)
    arr = (be.BwdTerm * n)()
    gus = [torch.empty(shape, dtype=torch.float32, device=device) for _ in specs]
    keep = []
    for i, s_ in enumerate(specs):
        arr[i].linop, arr[i].prox, arr[i].alpha = s_["linop"], s_["prox"], float(s_.get("alpha", 1.0))
        for name in ("lam", "v", "gv", "gu_new"):
            t = s_.get(name)
            if t is not None:
                t = t.contiguous()
                keep.append(t)
            setattr(arr[i], name, None if t is None else t.data_ptr())
        arr[i].gu = gus[i].data_ptr()
    gx = torch.empty(shape, dtype=torch.float32, device=device)
    glam = torch.empty(n, B, dtype=torch.float32, device=device)
    be.lib().call("dpx_admm_zupdate_bwd", ptr(gx), arr, n, ptr(glam), B, C, H, W, ptr(_bwd_ws(B, C, H, W, device)), be.stream())
    return gx, gus, [glam[i] for i in range(n)]


def admm_solve_rho_grad(g_rhs, x, linops):
    B, C, H, W = _shape4(x)
    n = len(linops)
    codes = (ctypes.c_int * max(n, 1))(*linops)
    out = torch.empty(B, dtype=torch.float32, device=x.device)
    be.lib().call("dpx_admm_solve_rho_grad", ptr(g_rhs.contiguous()), ptr(x.contiguous()), codes, n, ptr(out), B, C, H, W,
                  ptr(_bwd_ws(B, C, H, W, x.device)), be.stream())
    return out


def admm_rhs_bwd(g, rhs, rho, linops, want_v=True, want_u=True):
    B, C, H, W = _shape4(g)
    n = len(linops)
    codes = (ctypes.c_int * n)(*linops)
    gv = [torch.empty_like(g) if want_v else None for _ in range(n)]
    gu = [torch.empty_like(g) if want_u else None for _ in range(n)]
    pv = (c_void_p * n)(*[None if t is None else t.data_ptr() for t in gv])
    pu = (c_void_p * n)(*[None if t is None else t.data_ptr() for t in gu])
    grho = torch.empty(B, dtype=torch.float32, device=g.device)
    be.lib().call("dpx_admm_rhs_bwd", ptr(g.contiguous()), ptr(rhs), ptr(rho.contiguous()), codes, n, pv, pu, ptr(grho), B, C, H, W,
                  ptr(_bwd_ws(B, C, H, W, g.device)), be.stream())
    return gv, gu, grho


def make_terms(specs):
    """specs: list of dict(linop, prox, alpha, lam[B] tensor or None, v, u) -> (ctypes array, keepalive)."""
    n = len(specs)
    if n > be.MAX_TERMS:
        raise be.DpxError(f"at most {be.MAX_TERMS} fused terms")
    arr = (Term * max(n, 1))()
    for i, s in enumerate(specs):
        arr[i].linop, arr[i].prox, arr[i].alpha = s["linop"], s["prox"], float(s.get("alpha", 1.0))
        arr[i].lam = None if s.get("lam") is None else s["lam"].data_ptr()
        arr[i].v, arr[i].u = s["v"].data_ptr(), s["u"].data_ptr()
        arr[i].u_out = None if s.get("u_out") is None else s["u_out"].data_ptr()
    return arr


def admm_rhs(rhs, ktb, rho, term_arr, nterms):
    B, C, H, W = _shape4(rhs)
    be.lib().call("dpx_admm_rhs", ptr(rhs), ptr(ktb), ptr(rho), term_arr, nterms, B, C, H, W, be.stream())
    return rhs


def split_rhs(rhs, ktb, x, rho, term_arr, nterms, mode):
    """rhs = ktb + rho sum_i K_i^T (x - K_i^T q_i): mode 0 Pock-Chambolle (q = z), 1 linearised ADMM (q = K x - v + u)"""
    B, C, H, W = _shape4(rhs)
    be.lib().call("dpx_split_rhs", ptr(rhs), ptr(ktb), ptr(x), ptr(rho), term_arr, nterms, int(mode), B, C, H, W, be.stream())
    return rhs


def pc_dual(xbar, term_arr, nterms):
    B, C, H, W = _shape4(xbar)
    be.lib().call("dpx_pc_dual", ptr(xbar), term_arr, nterms, B, C, H, W, be.stream())


def admm_pnp_iter(x, rhs, term_arr, nterms, ext, v_new, rho, sigma, spec_add, dd, eps, net):
    """one plug-and-play ADMM iteration in one C call (dpx_admm_pnp_iter); `net`: the FFDNet module of term `ext`"""
    B, C, H, W = _shape4(x)
    L = be.lib()
    mode = be.FFDNET_MODES[net.compute_mode]
    if mode in (3, 4):
        be.note_f16_launch()
    Bn = B if net.in_nc == C else B * C
    if mode == 0:
        packed = net.packed()
        ws = workspace("ffdnet", L.query("dpx_ffdnet_ws_bytes", Bn, net.in_nc, net.nc, H, W), x.device)
    else:
        packed = net.packed_bf16(mode)
        ws = workspace("ffdnet_bf16", L.query("dpx_ffdnet_bf16_ws_bytes", Bn, net.in_nc, net.nc, H, W), x.device)
    L.call("dpx_admm_pnp_iter", ptr(x), ptr(rhs), term_arr, nterms, int(ext), ptr(v_new), ptr(rho), ptr(sigma), ptr(spec_add), ptr(dd),
           c_float(eps), ptr(packed), net.in_nc, net.nc, net.nb, mode, B, C, H, W, ptr(fft_table(H, W, x.device)),
           ptr(spectrum_ws(B * C, H, W, x.device)), ptr(ws), be.stream())


class CgPnpIter:
    """One plug-and-play iteration with the masked-Fourier CG x-update in one C call (dpx_admm_cg_pnp_iter), set up once per run: everything
    that does not change between iterations (workspaces, packed weights, the table, the mask) is resolved here, the per-iteration call
    passes pointers only.  `net`: the gray FFDNet of term `ext`; x: [B, 1, H, W]."""

    def __init__(self, x, rhs, ktb, term_arr, nterms, ext, mask, n_identity, rtol, max_iters, net):
        B, C, H, W = _shape4(x)
        assert C == 1 and net.in_nc == 1
        L = be.lib()
        self.L = L
        self.mode = be.FFDNET_MODES[net.compute_mode]
        mask = mask.to(device=x.device, dtype=torch.float32).contiguous()
        mimg = B if mask.numel() == x.numel() else 1
        assert mask.numel() == mimg * H * W
        if self.mode == 0:
            packed = net.packed()
            ffd_ws = workspace("ffdnet", L.query("dpx_ffdnet_ws_bytes", B, net.in_nc, net.nc, H, W), x.device)
        else:
            packed = net.packed_bf16(self.mode)
            ffd_ws = workspace("ffdnet_bf16", L.query("dpx_ffdnet_bf16_ws_bytes", B, net.in_nc, net.nc, H, W), x.device)
        cg_ws = workspace("cg_masked_fft", L.query("dpx_cg_masked_fft_ws_bytes", B, H, W, mimg), x.device)
        self.keep = (mask, packed, ffd_ws, cg_ws, ktb, rhs, fft_table(H, W, x.device))
        self.fixed = (ptr(mask), mimg, c_float(n_identity), c_float(rtol), int(max_iters), ptr(packed), net.in_nc, net.nc, net.nb, self.mode, B, H, W,
                      ptr(self.keep[6]), ptr(cg_ws), ptr(ffd_ws))
        self.head = (ptr(rhs), ptr(ktb), term_arr, nterms, int(ext))
        self.folds = bool(L.query("dpx_admm_cg_pnp_iter_folds", self.mode, B))
        self.ready = False        # the previous call prepared this call's right-hand side

    def __call__(self, x, v_new, rho, sigma, rho_next=None, x_next=None, cg_hint=-1):
        """x (written), v_new (receives the denoised image); returns the CG exit iteration.  rho_next / x_next (self.folds only): the pass
        behind the denoiser also prepares the next call's right-hand side and CG start state (x_next zeroed: the next call's x) -- the next
        call then skips its rhs stage by itself"""
        if self.mode in (3, 4):
            be.note_f16_launch()
        L = self.L
        ready, self.ready = self.ready, rho_next is not None
        n = L.query("dpx_admm_cg_pnp_iter", ptr(x), *self.head, ptr(v_new), ptr(rho), ptr(sigma), *self.fixed, ptr(rho_next), ptr(x_next), int(ready),
                    int(cg_hint), be.stream())
        if n < 0:
            raise be.DpxError(f"dpx_admm_cg_pnp_iter failed ({n}): {L.cdll.dpx_last_error().decode()}")
        return int(n)


def admm_zupdate(x, term_arr, nterms):
    B, C, H, W = _shape4(x)
    be.lib().call("dpx_admm_zupdate", ptr(x), term_arr, nterms, B, C, H, W, be.stream())


def admm_zupdate_rhs(x, term_arr, nterms, rhs, rho_next, dual=True, emit_v=True, ktb=None):
    """z / dual update of this iteration and the right-hand side of the next one in one pass (dpx_admm_zupdate_rhs: every term's dual is
    double-buffered, u_out != u; emit_v=False: v is not stored -- the loop's last stage must be admm_zupdate)"""
    B, C, H, W = _shape4(x)
    be.lib().call("dpx_admm_zupdate_rhs", ptr(x), term_arr, nterms, ptr(rhs), ptr(ktb), ptr(rho_next), int(bool(dual)), int(bool(emit_v)), B, C, H, W, be.stream())


def iter_supported(H, W, term_arr, nterms):
    return bool(be.lib().query("dpx_admm_iter_supported", H, W, term_arr, nterms))


def spectrum_buffer(P, H, W, device):
    """one half-spectrum buffer (dpx_spectrum_bytes covers two)"""
    return _bytes(be.lib().query("dpx_spectrum_bytes", P, H, W) // 2, device)


def rfft_rows(x, spec):
    B, C, H, W = _shape4(x)
    be.lib().call("dpx_rfft_rows", ptr(x), ptr(spec), B, C, H, W, ptr(fft_table(H, W, x.device)), be.stream())


def iter_cols(spec_in, spec_out, spec_add, dd, rho, eps, shape, device):
    B, C, H, W = shape
    be.lib().call("dpx_admm_iter_cols", ptr(spec_in), ptr(spec_out), ptr(spec_add), ptr(dd), ptr(rho), c_float(eps),
                  B, C, H, W, ptr(fft_table(H, W, device)), be.stream())


def iter_rows(spec_in, spec_out, term_arr, nterms, rho_next, x_out, emit_v, shape, device):
    B, C, H, W = shape
    be.lib().call("dpx_admm_iter_rows", ptr(spec_in), ptr(spec_out), term_arr, nterms, ptr(rho_next), ptr(x_out),
                  int(bool(emit_v)), B, C, H, W, ptr(fft_table(H, W, device)), be.stream())


def admm_seed_rows(spec, rho, term_arr, nterms, shape, device, fresh_x=None, stream=None):
    """spec = row transform of rho_b sum_i K_i^T (v_i - u_i): the seed of admm_run in one pass.  ``fresh_x``: the state is
    ADMM.initialize(fresh_x) untouched (v_i = K_i x0, u_i = 0) -- the pass then reads x0 alone (bit-identical result).
    stream: a raw stream handle (default: the current stream)"""
    B, C, H, W = shape
    st = be.stream() if stream is None else stream
    if fresh_x is not None:
        be.lib().call("dpx_admm_seed_rows_fresh", ptr(spec), ptr(rho), ptr(fresh_x), term_arr, nterms, B, C, H, W, ptr(fft_table(H, W, device)), st)
        return spec
    be.lib().call("dpx_admm_seed_rows", ptr(spec), ptr(rho), term_arr, nterms, B, C, H, W, ptr(fft_table(H, W, device)), st)
    return spec


def stream_fork(frm, to):
    """every raw stream of ``to`` waits for what has been issued on ``frm`` (dpx_stream_fork; no-op on the CPU emulator)"""
    if be.host_mode():
        return
    arr = (c_void_p * len(to))(*to)
    be.lib().call("dpx_stream_fork", c_void_p(frm), arr, len(to))


def stream_join(into, frm):
    """``into`` waits for what has been issued on every raw stream of ``frm``"""
    if be.host_mode():
        return
    arr = (c_void_p * len(frm))(*frm)
    be.lib().call("dpx_stream_join", c_void_p(into), arr, len(frm))


def admm_run(spec_a, spec_b, spec_add, dd, term_arr, nterms, rho_tab, lam_tabs, eps, it0, n_iters, total, x_out, emit_last,
             shape, device):
    """n_iters fused iterations on the C side; returns the u-buffer parity (0: terms[i].u current, 1: u_out).
    emit_last: 0 nothing / 1 x and v (and the duals) / 2 x alone when the call ends the solve (v and u are then NOT updated)"""
    B, C, H, W = shape
    lt = (c_void_p * nterms)(*[None if t is None else t.data_ptr() for t in lam_tabs])
    L = be.lib()
    rc = L.query("dpx_admm_run", ptr(spec_a), ptr(spec_b), ptr(spec_add), ptr(dd), term_arr, nterms, ptr(rho_tab), lt,
                 c_float(eps), it0, n_iters, total, ptr(x_out), int(emit_last), B, C, H, W,
                 ptr(fft_table(H, W, device)), be.stream())
    if rc < 0:
        raise be.DpxError(f"dpx_admm_run failed ({rc}): {L.cdll.dpx_last_error().decode()}")
    return rc


def _addr(t):
    """device address of a tensor, or the address itself"""
    return t if isinstance(t, int) else t.data_ptr()


def admm_run_chains(chains, dd, nterms, eps, it0, n_iters, total, emit_last, shape, device):
    """dpx_admm_run for several sub-batch chains at once.  chains: dicts with spec_a, spec_b, spec_add (tensor / None), terms (ctypes
    array), rho_tab, lam_tabs (tensors), x_out, B, stream (raw handle), optionally seed (0 / 1 / 2) and seed_x0.  Returns the dual-buffer parity."""
    _, C, H, W = shape
    arr = (be.Chain * len(chains))()
    keep = []
    for i, ch in enumerate(chains):
        lt = (c_void_p * nterms)(*[None if t is None else t.data_ptr() for t in ch["lam_tabs"]])
        keep.append(lt)
        arr[i].spec_a, arr[i].spec_b = _addr(ch["spec_a"]), _addr(ch["spec_b"])
        arr[i].spec_add = None if ch["spec_add"] is None else _addr(ch["spec_add"])
        arr[i].terms = ctypes.cast(ch["terms"], ctypes.POINTER(Term))
        arr[i].rho_tab = ch["rho_tab"].data_ptr()
        arr[i].lam_tabs = ctypes.cast(lt, ctypes.POINTER(c_void_p))
        arr[i].x_out = _addr(ch["x_out"])
        arr[i].B = int(ch["B"])
        arr[i].stream = ch["stream"]
        arr[i].seed = int(ch.get("seed", 0))
        arr[i].seed_x0 = None if ch.get("seed_x0") is None else _addr(ch["seed_x0"])
    L = be.lib()
    rc = L.query("dpx_admm_run_chains", arr, len(chains), ptr(dd), nterms, c_float(eps), it0, n_iters, total, int(emit_last), C, H, W,
                 ptr(fft_table(H, W, device)))
    if rc < 0:
        raise be.DpxError(f"dpx_admm_run_chains failed ({rc}): {L.cdll.dpx_last_error().decode()}")
    return rc


def mul(x, w):
    """x * w with w one image ([1,C,H,W] or [C,H,W]) or a batch of them"""
    require(x, what="mul input")
    B = int(x.shape[0])
    npb = x.numel() // B
    w = w.to(device=x.device, dtype=torch.float32).contiguous()
    if w.numel() == npb:
        wimg = 1
    elif w.numel() == x.numel():
        wimg = B
    else:
        raise be.DpxError(f"mul: weight {tuple(w.shape)} matches neither one image nor the batch {tuple(x.shape)}")
    out = torch.empty_like(x)
    be.lib().call("dpx_mul", ptr(x), ptr(w), ptr(out), B, npb, wimg, be.stream())
    return out


def wss_prox(v, ktb, diag, lam):
    """(ktb + lam v) / (diag + lam) with per-image lam; ktb / diag one image or a batch"""
    require(v, what="wss_prox input")
    B = int(v.shape[0])
    npb = v.numel() // B
    ktb = ktb.to(device=v.device, dtype=torch.float32).contiguous()
    diag = diag.to(device=v.device, dtype=torch.float32).contiguous()

    def images(t, what):
        if t.numel() == npb:
            return 1
        if t.numel() == v.numel():
            return B
        raise be.DpxError(f"wss_prox: {what} {tuple(t.shape)} matches neither one image nor the batch {tuple(v.shape)}")
    out = torch.empty_like(v)
    be.lib().call("dpx_wss_prox", ptr(v), ptr(ktb), images(ktb, "Ktb"), ptr(diag), images(diag, "diag"), ptr(as_batch_vec(lam, B, v.device)),
                  ptr(out), B, npb, be.stream())
    return out


def mul_color(x, srf, transpose=False):
    """channel mixing by srf [C, C2]: forward srf.T @ x (C -> C2 channels), adjoint srf @ x (C2 -> C)"""
    require(x, what="mul_color input")
    B, Cx, H, W = _shape4(x)
    srf = srf.to(device=x.device, dtype=torch.float32).contiguous()
    C, C2 = int(srf.shape[0]), int(srf.shape[1])
    if Cx != (C2 if transpose else C):
        raise be.DpxError(f"mul_color: input has {Cx} channels, srf is {C}x{C2} (transpose={transpose})")
    out = torch.empty(B, C if transpose else C2, H, W, dtype=torch.float32, device=x.device)
    be.lib().call("dpx_mul_color", ptr(x), ptr(srf), ptr(out), int(bool(transpose)), B, C, C2, H * W, be.stream())
    return out


def upsample_zero(y, sf):
    require(y, what="upsample input")
    B, C, h, w = _shape4(y)
    out = torch.empty(B, C, h * sf, w * sf, dtype=torch.float32, device=y.device)
    be.lib().call("dpx_upsample_zero", ptr(y), ptr(out), int(sf), B * C, h, w, be.stream())
    return out


def cplx_mul(a, b, conj_a=False):
    """(conj) a * b, complex64; a is one image [1,...] shared by the batch or a batch like b"""
    require(b, dtype=torch.complex64, what="cplx_mul b")
    a = a.to(torch.complex64).contiguous()
    B = int(b.shape[0])
    npb = b.numel() // B
    a = a.expand(*([1] * (b.ndim - a.ndim)), *a.shape) if a.ndim < b.ndim else a
    if a.numel() == npb:
        aimg = 1
    elif a.numel() == b.numel():
        aimg = B
    else:
        a = a.expand_as(b).contiguous()
        aimg = B
    out = torch.empty_like(b)
    be.lib().call("dpx_cplx_mul", ptr(out), ptr(a), ptr(b), int(bool(conj_a)), B, npb, aimg, be.stream())
    return out


def sisr_update(FR, FB, lam, I, sf):
    """FX of sr.py:66-72 from FR [B,C,H,W] and the kernel spectrum FB ([1|B, 1|C, H, W])"""
    require(FR, dtype=torch.complex64, what="sisr FR")
    B, C, H, W = _shape4(FR)
    FB = FB.to(torch.complex64).contiguous()
    planes = FB.numel() // (H * W)
    if planes not in (1, C, B * C):
        FB = FB.expand(B, C, H, W).contiguous()
        planes = B * C
    buf = torch.empty(2, B, C, H, W, dtype=torch.complex64, device=FR.device)
    buf[0].copy_(FR)
    be.lib().call("dpx_sisr_update", ptr(buf), ptr(FB), planes, ptr(as_batch_vec(lam, B, FR.device)), float(I), int(sf), B, C, H, W,
                  be.stream())
    return buf[1]


def conv_pack(w, b, taps):
    """w [cout, cin, taps] float32 device tensor (+ optional bias) -> packed blob for conv2d"""
    require(w, what="conv weight")
    cout, cin = int(w.shape[0]), int(w.shape[1])
    L = be.lib()
    blob = torch.empty(L.query("dpx_conv_packed_bytes", cin, cout, taps), dtype=torch.uint8, device=w.device)
    L.call("dpx_conv_pack", ptr(blob), ptr(w), ptr(b), cin, cout, taps, be.stream())
    return blob


def conv2d(x, packed, cout, taps, relu=False, res=None, dilation=1):
    """stride-1 3x3 (pad = dilation) / 1x1 convolution on the fp32 matrix cores; optional ReLU or residual add"""
    require(x, what="conv input")
    B, C, H, W = _shape4(x)
    out = torch.empty(B, cout, H, W, dtype=torch.float32, device=x.device)
    if res is not None:
        require(res, what="conv residual")
    be.lib().call("dpx_conv2d", ptr(x), ptr(out), ptr(packed), ptr(res), int(bool(relu)), C, cout, taps, int(dilation), B, H, W, be.stream())
    return out


def conv2d_leaky(x, packed, cout, neg_slope=0.2):
    """3x3 / pad 1 convolution + bias + LeakyReLU(neg_slope) fused (the U-Net denoiser's ConvLayer)"""
    require(x, what="conv input")
    B, C, H, W = _shape4(x)
    out = torch.empty(B, cout, H, W, dtype=torch.float32, device=x.device)
    be.lib().call("dpx_conv2d_leaky", ptr(x), ptr(out), ptr(packed), float(neg_slope), C, cout, 9, B, H, W, be.stream())
    return out


def leaky_relu_bwd(y, g, neg_slope=0.2):
    """g * (y > 0 ? 1 : neg_slope): backward of the fused LeakyReLU from the layer's saved output"""
    require(y, what="activation")
    require(g, what="gradient")
    out = torch.empty_like(g)
    be.lib().call("dpx_leaky_relu_bwd", ptr(y), ptr(g), ptr(out), y.numel(), float(neg_slope), be.stream())
    return out


def maxpool2(x):
    """MaxPool2d(2) (floor)"""
    require(x, what="maxpool input")
    B, C, H, W = _shape4(x)
    y = torch.empty(B, C, H // 2, W // 2, dtype=torch.float32, device=x.device)
    be.lib().call("dpx_maxpool2", ptr(x), ptr(y), B, C, H, W, be.stream())
    return y


def maxpool2_bwd(x, gy):
    require(x, what="maxpool input")
    require(gy, what="maxpool output gradient")
    B, C, H, W = _shape4(x)
    gx = torch.empty_like(x)
    be.lib().call("dpx_maxpool2_bwd", ptr(x), ptr(gy), ptr(gx), B, C, H, W, be.stream())
    return gx


def concat_skip_upsampled(skip, low):
    """torch.cat([skip, pad(upsample_x2_bilinear_align_corners(low))], dim=1) without the intermediates: the interpolated,
    zero-padded tensor is written straight into its channel slice (reference models/unet/unet.py:96-117)"""
    require(skip, what="skip tensor")
    require(low, what="low-resolution tensor")
    B, C2, H, W = _shape4(skip)
    _, C1, h, w = _shape4(low)
    out = torch.empty(B, C2 + C1, H, W, dtype=torch.float32, device=skip.device)
    L = be.lib()
    L.call("dpx_copy_channels", ptr(skip), ptr(out), 1, B, C2, H, W, C2 + C1, 0, be.stream())
    L.call("dpx_upsample2_into", ptr(low), ptr(out), B, C1, h, w, C2 + C1, C2, H, W, be.stream())
    return out


def concat_skip_upsampled_bwd(g, C2, low_shape):
    """gradients of concat_skip_upsampled w.r.t. (skip, low)"""
    require(g, what="concat gradient")
    B, Ct, H, W = _shape4(g)
    C1, (h, w) = Ct - C2, low_shape
    gskip = torch.empty(B, C2, H, W, dtype=torch.float32, device=g.device)
    glow = torch.empty(B, C1, h, w, dtype=torch.float32, device=g.device)
    L = be.lib()
    L.call("dpx_copy_channels", ptr(g), ptr(gskip), 0, B, C2, H, W, Ct, 0, be.stream())
    L.call("dpx_upsample2_into_bwd", ptr(g), ptr(glow), B, C1, h, w, Ct, C2, H, W, be.stream())
    return gskip, glow


def conv2d_wgrad(g, a, taps, dilation=1, want_bias=False):
    """weight (and bias) gradient of one conv2d layer: g [B,cout,H,W] gradient of the pre-activation output, a [B,cin,H,W] the
    layer's input -> gw [cout, cin, taps] (, gb [cout])"""
    require(g, what="conv output gradient")
    require(a, what="conv input")
    B, cout, H, W = _shape4(g)
    cin = int(a.shape[1])
    L = be.lib()
    gw = torch.empty(cout, cin, taps, dtype=torch.float32, device=g.device)
    gb = torch.empty(cout, dtype=torch.float32, device=g.device) if want_bias else None
    ws = workspace("conv_wgrad", L.query("dpx_conv2d_wgrad_ws_bytes", cin, cout, taps, B, H, W), g.device)
    L.call("dpx_conv2d_wgrad", ptr(g), ptr(a), ptr(gw), ptr(gb), cin, cout, taps, int(dilation), B, H, W, ptr(ws), be.stream())
    return gw, gb


def space_to_depth(x):
    require(x, what="space_to_depth input")
    B, C, H, W = _shape4(x)
    y = torch.empty(B, 4 * C, H // 2, W // 2, dtype=torch.float32, device=x.device)
    be.lib().call("dpx_space_to_depth", ptr(x), ptr(y), B, C, H, W, be.stream())
    return y


def depth_to_space(x):
    require(x, what="depth_to_space input")
    B, C4, H, W = _shape4(x)
    y = torch.empty(B, C4 // 4, 2 * H, 2 * W, dtype=torch.float32, device=x.device)
    be.lib().call("dpx_depth_to_space", ptr(x), ptr(y), B, C4 // 4, H, W, be.stream())
    return y


def cplx_scale(a, w):
    """w * a with a complex64 and w a real weight (one image or a batch)"""
    require(a, dtype=torch.complex64, what="cplx_scale input")
    B = int(a.shape[0])
    npb = a.numel() // B
    w = w.to(device=a.device, dtype=torch.float32).contiguous()
    if w.numel() == npb:
        wimg = 1
    elif w.numel() == a.numel():
        wimg = B
    else:
        raise be.DpxError(f"cplx_scale: weight {tuple(w.shape)} matches neither one image nor the batch {tuple(a.shape)}")
    out = torch.empty_like(a)
    be.lib().call("dpx_cplx_scale", ptr(out), ptr(a), ptr(w), B, npb, wimg, be.stream())
    return out


def clincomb(terms, out_complex=True):
    """sum_i coef_i * x_i over up to 4 real-fp32 / complex64 tensors of one shape; complex64 result, or its real part
    as fp32 (out_complex=False).  coef_i are python floats."""
    xs = [t[1].contiguous() for t in terms]
    ref = xs[0]
    if not 1 <= len(xs) <= 4:
        raise be.DpxError("clincomb: 1..4 operands")
    for x in xs:
        if x.dtype not in (torch.float32, torch.complex64):
            raise be.DpxError(f"clincomb operand must be float32 or complex64, got {x.dtype}")
        require(x, dtype=None, what="clincomb operand")
        if x.shape != ref.shape:
            raise be.DpxError(f"clincomb: shape mismatch {tuple(x.shape)} vs {tuple(ref.shape)}")
    res = torch.empty(ref.shape, dtype=torch.complex64 if out_complex else torch.float32, device=ref.device)
    if ref.numel() == 0:
        return res
    n = len(xs)
    px = (c_void_p * n)(*[x.data_ptr() for x in xs])
    cx = (ctypes.c_int * n)(*[int(x.is_complex()) for x in xs])
    cf = (c_float * n)(*[float(t[0]) for t in terms])
    be.lib().call("dpx_cplx_lincomb", ptr(res), int(bool(out_complex)), n, px, cx, cf, ref.numel(), be.stream())
    return res


def csmri_update(z, y, mask, lam, num_psi):
    """in place on the complex64 spectrum z [B,C,H,W]: z[mask] = ((lam z + y) / (1 + lam num_psi))[mask]"""
    require(z, dtype=torch.complex64, what="csmri spectrum")
    B = int(z.shape[0])
    npi = z.numel() // B
    y = y.to(torch.complex64).contiguous()
    if y.shape != z.shape:
        raise be.DpxError(f"csmri: y {tuple(y.shape)} does not match the iterate {tuple(z.shape)}")
    mk = (mask != 0).to(torch.uint8).contiguous()
    if mk.numel() == z.numel():
        mimg = B
    elif mk.numel() == npi:
        mimg = 1
    else:
        raise be.DpxError(f"csmri: mask {tuple(mask.shape)} matches neither one image nor the batch {tuple(z.shape)}")
    lam_v = as_batch_vec(lam, B, z.device)
    be.lib().call("dpx_csmri_update", ptr(z), ptr(y), ptr(mk), mimg, ptr(lam_v), float(num_psi), B, npi, be.stream())
    return z


def otf_grad(A, X, Y, O):
    """dL/dO of the Fourier x-update (dpx_otf_grad): A, X, Y complex64 [B,C,H,W] (Y may be None), O complex64 [1,C,H,W] -> [1,C,H,W]"""
    B, C, H, W = (int(v) for v in A.shape)
    for t, what in ((A, "A"), (O, "O")) + (((X, "X"),) if X is not None else ()) + (((Y, "Y"),) if Y is not None else ()):
        require(t, dtype=torch.complex64, what=f"otf_grad {what}")
    G = torch.empty_like(O)
    be.lib().call("dpx_otf_grad", ptr(A), ptr(X), ptr(Y), ptr(O), ptr(G), B, C, H, W, 0, be.stream())
    return G


def cfft2(x, inverse=False, centred=True, ortho=True):
    """complex64 2-D FFT over the last two dims (hand-written kernels; centring shifts and normalisation fused)"""
    if not x.is_complex():
        x = clincomb([(1.0, x.float())], out_complex=True)
    x = x.to(torch.complex64).contiguous()
    require(x, dtype=torch.complex64, what="cfft2 input")
    H, W = int(x.shape[-2]), int(x.shape[-1])
    P = x.numel() // (H * W)
    out = torch.empty_like(x)
    be.lib().call("dpx_cfft2", ptr(x), ptr(out), int(bool(inverse)), int(bool(centred)), int(bool(ortho)), P, H, W,
                  ptr(fft_table(H, W, x.device)), be.stream())
    return out
