"""``ProxFn`` -- proximal-operator protocol (reference dprox/proxfn/base.py:30-108).

    prox_f(v, lam) = argmin_x f(x) + 1/(2 lam) ||x - v||^2

A term of the objective is ``alpha * f(beta * K x - b)``: subclasses implement ``_prox(v, lam)`` for the
bare f; ``prox`` applies the scaled / affine / translated wrappers (base.py:12-27):

    prox(v, lam) = 1/beta * _prox(beta * (v - off), beta^2 * alpha * lam) + off ,   off = -K(0)

Built-in terms set ``hip_kind`` and run as ONE fused HIP pass (``dpx_prox``); plugin subclasses written
with torch ops keep working through the generic wrapper.  The offset is evaluated once and cached (the
reference re-runs the whole DAG, FFTs included, on every call) and invalidated when a ``Placeholder``
of the term is re-assigned.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import _ops as ops
from ..linop import CompGraph, LinOp, Placeholder
from ..utils import to_torch_tensor


def exists(x):
    return x is not None


class ProxFn(nn.Module):
    hip_kind = None          # DPX_PROX_* code when the bare prox is a built-in HIP kernel

    def __init__(self, linop: LinOp, alpha=1, beta=1):
        super().__init__()
        self.linop = linop
        self.alpha = alpha
        self.beta = beta
        self.step = 0
        self.dag = CompGraph(linop, zero_out_constant=True)
        self._offset_cache = None

    # ---- offset ---------------------------------------------------------------------------------
    def _offset_key(self):
        vs = self.linop.variables
        shapes = tuple((tuple(v._value.shape), str(v._value.dtype)) if v._value is not None else None for v in vs)
        vers = tuple(getattr(c, "_version", 0) for c in self.linop.constants)
        return (shapes, vers, str(self.linop.device))

    def _compute_offset(self):
        if not self.linop.constants:
            return None
        off = self.linop.offset
        return None if off is None else ops.lincomb([(-1.0, off)])

    @property
    def offset(self):
        """-K(0); ``None`` when the term has no constant part (zero offset)."""
        key = self._offset_key()
        if self._offset_cache is None or self._offset_cache[0] != key:
            self._offset_cache = (key, self._compute_offset())
        return self._offset_cache[1]

    def unwrap(self, value):
        if isinstance(value, Placeholder):
            return value.value
        t = to_torch_tensor(value, batch=True)
        dt = torch.complex64 if t.is_complex() else torch.float32
        return t.to(device=self.linop.device, dtype=dt)

    # ---- prox -----------------------------------------------------------------------------------
    def eval(self, v=None):
        if v is None:                      # nn.Module.eval()
            return super().eval()
        return NotImplementedError

    def _prox(self, v, lam):
        return NotImplementedError

    def prox(self, v, lam):
        """v: [B,C,H,W]; lam: 0-d or [B]"""
        off = self.offset
        if off is not None and off.shape != v.shape:
            off = off.expand_as(v).contiguous()
        if self.hip_kind is not None and self.beta == 1 and v.dtype == torch.float32:
            return ops.prox(self.hip_kind, v, lam, float(self.alpha), off)
        if lam.ndim == 1:
            lam = lam.view(lam.shape[0], 1, 1, 1)
        b = self.beta
        t = v
        if off is not None or b != 1:
            t = ops.lincomb([(float(b), v)] + ([(-float(b), off)] if off is not None else []))
        p = self._prox(t, (b * b * lam) * self.alpha)
        if off is None and b == 1:
            return p
        return ops.lincomb([(1.0 / b, p.contiguous())] + ([(1.0, off)] if off is not None else []))

    def convex_conjugate_prox(self, v, lam):
        """Moreau: v - prox(v / lam, lam)"""
        lam4 = lam.view(lam.shape[0], 1, 1, 1) if lam.ndim == 1 else lam
        p = self.prox((v / lam4).contiguous(), lam)
        return ops.lincomb([(1.0, v), (-1.0, p)])

    # ---- composition ----------------------------------------------------------------------------
    def __mul__(self, other):
        if np.isscalar(other) and other > 0:
            self.alpha = other
            return self
        return TypeError("Can only multiply by a positive scalar.")

    __rmul__ = __mul__

    def __add__(self, other):
        if isinstance(other, ProxFn):
            return [self, other]
        if type(other) == list:
            return [self] + other
        return NotImplemented

    def __radd__(self, other):
        if type(other) == list:
            return other + [self]
        return NotImplemented

    def __str__(self):
        return f"{self.__class__.__name__}"
