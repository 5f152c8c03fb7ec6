"""``sisr`` -- single-image super-resolution data term with the closed-form x-update of Zhao et al.
(reference dprox/proxfn/fast/sr.py:45-126): y = (x * k) downsampled by ``sf``.

    FR  = conj(FB) F(S^T y) + F(lam v)
    FX  = (FR - conj(FB) * mean_aliases(FB FR) / (mean_aliases(|FB|^2) + I lam)) / (I lam + 1e-9)
    x   = real(F^-1 FX)

The transforms are ``dpx_cfft2`` (hand-written complex 2-D FFT), the aliasing algebra one kernel (``dpx_sisr_update``).
"""
import numpy as np
import torch

from ... import _ops as ops
from ..quadratic import ext_sum_squares


def p2o(psf, shape):
    """kernel [N,C,kh,kw] -> zero-padded to `shape`, centre rolled to (0,0) (sr.py:95-114); the transform runs on the device"""
    psf = np.asarray(psf, dtype=np.float32)
    otf = np.zeros(psf.shape[:-2] + tuple(shape), dtype=np.float32)
    otf[..., :psf.shape[-2], :psf.shape[-1]] = psf
    for axis, axis_size in enumerate(psf.shape[-2:]):
        otf = np.roll(otf, -int(axis_size / 2), axis=axis + otf.ndim - 2)
    return otf


class sisr(ext_sum_squares):
    def __init__(self, linop, y, kernel, sf):
        super().__init__(linop)
        self.sf = sf
        self.y = y
        self.k = kernel
        self._cache = None

    def reload(self):
        k = self.unwrap(self.k)
        y = self.unwrap(self.y).float().contiguous()
        key = (k.data_ptr(), k._version, y.data_ptr(), y._version, tuple(y.shape), str(y.device))
        if self._cache is not None and self._cache[0] == key:
            return self._cache[1]
        h, w = y.shape[-2:]
        H, W = h * self.sf, w * self.sf
        padded = torch.from_numpy(p2o(k.detach().cpu().numpy(), (H, W))).to(y.device)
        FB = ops.cfft2(padded.reshape(-1, 1, H, W).contiguous(), inverse=False, centred=False, ortho=False).reshape(*padded.shape)
        STy = ops.upsample_zero(y, self.sf)
        FBFy = ops.cplx_mul(FB, ops.cfft2(STy, inverse=False, centred=False, ortho=False), conj_a=True)
        self._cache = (key, (FB, FBFy))
        return self._cache[1]

    def _prox(self, v, lam, I):
        FB, FBFy = self.reload()
        v = v.contiguous().float()
        B = v.shape[0]
        lam_b = ops.as_batch_vec(lam, B, v.device)
        lv = ops.lincomb([(lam_b, v)])
        FR = ops.clincomb([(1.0, FBFy.expand_as(v).contiguous() if FBFy.shape != v.shape else FBFy),
                           (1.0, ops.cfft2(lv, inverse=False, centred=False, ortho=False))], out_complex=True)
        FX = ops.sisr_update(FR, FB, lam_b, float(I), self.sf)
        X = ops.cfft2(FX.contiguous(), inverse=True, centred=False, ortho=False)
        return ops.clincomb([(1.0, X)], out_complex=False)
