"""``csmri`` -- compressed-sensing MRI data term with a closed-form x-update
(reference dprox/proxfn/fast/csmri.py:8-25):

    z = fft2(v);   z[mask] = ((lam z + y) / (1 + lam num_psi))[mask];   return ifft2(z)

with the centred orthonormal transforms of ``dprox.utils`` -- here ``dpx_cfft2`` (hand-written complex 2-D FFT,
shifts and scaling fused) around ``dpx_csmri_update`` (the masked Fourier-domain update, in place).
"""
import torch

from ... import _ops as ops
from ..quadratic import ext_sum_squares


class csmri(ext_sum_squares):
    def __init__(self, linop, mask, y):
        super().__init__(linop)
        self.mask = mask
        self.y = y

    def _prox(self, v, lam, num_psi):
        y = self.unwrap(self.y)
        mask = self.unwrap(self.mask)
        z = ops.cfft2(v, inverse=False, centred=True, ortho=True)          # complex64, also for a real iterate
        ops.csmri_update(z, y.to(z.device), mask.to(z.device), lam, num_psi)
        return ops.cfft2(z, inverse=True, centred=True, ortho=True)
