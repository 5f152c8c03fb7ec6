"""Separable proximal operators: ``norm1`` (soft threshold), ``norm2``, ``nonneg``
(reference dprox/proxfn/norm.py:6-27, dprox/proxfn/nonneg.py:6-11) -- built-in HIP kernels."""
import torch

from .. import _backend as be
from .. import _ops as ops
from .core import ProxFn


def soft_threshold(v, lam):
    """argmin_x lam |x|_1 + 0.5 (x - v)^2  =  sign(v) * max(|v| - lam, 0)"""
    B = v.shape[0]
    lam_t = lam if isinstance(lam, torch.Tensor) else torch.tensor(float(lam))
    if lam_t.numel() not in (1, B):
        raise ValueError("soft_threshold: lam must be a scalar or one value per image")
    return ops.prox(be.PROX_NORM1, v, lam_t, 1.0, None)


class norm1(ProxFn):
    hip_kind = be.PROX_NORM1

    def __init__(self, linop=None):
        super().__init__(linop)

    def _prox(self, v, lam):
        return soft_threshold(v, lam)


class norm2(ProxFn):
    hip_kind = be.PROX_SUMSQ

    def __init__(self, linop=None):
        super().__init__(linop)

    def _prox(self, v, lam):
        return ops.prox(be.PROX_SUMSQ, v, lam, 1.0, None)


class nonneg(ProxFn):
    hip_kind = be.PROX_NONNEG

    def __init__(self, linop=None):
        super().__init__(linop)

    def _prox(self, v, lam):
        if v.is_complex():
            raise RuntimeError("nonneg is undefined for complex iterates (torch.maximum does not support complex)")
        return ops.prox(be.PROX_NONNEG, v, None, 1.0, None)
