"""Proximal gradient descent (reference dprox/algo/pgd.py:8-54):
x <- prox_psi(x - rho * K^T (K x - b), lam), exactly one smooth and one proxable term."""
from typing import List

from .. import _ops as ops
from ..proxfn import ProxFn
from .driver import Algorithm


class ProximalGradientDescent(Algorithm):
    @classmethod
    def partition(cls, prox_fns: List[ProxFn]):
        if len(prox_fns) != 2:
            raise ValueError("Proximal gradient descent only supports two proximal functions for now.")
        omega_fns = [fn for fn in prox_fns if hasattr(fn, "grad")]
        psi_fns = [fn for fn in prox_fns if not any(fn is o for o in omega_fns)]
        if len(omega_fns) == 0:
            raise ValueError("Proximal gradient descent requires at least one proximal function is differentiable.")
        return psi_fns, omega_fns

    def __init__(self, psi_fns, omega_fns, *args, **kwargs):
        super().__init__(psi_fns, omega_fns)
        self.diff_fn = omega_fns[0]
        self.prox_fn = psi_fns[0]

    def _iter(self, state, rho, lam):
        x = state[0]
        g = self.diff_fn.grad(x)
        v = ops.lincomb([(1.0, x), (-rho if rho.ndim else -float(rho), g)])
        return [self.prox_fn.prox(v, lam[self.prox_fn])]

    def initialize(self, x0):
        return [x0]

    @property
    def state_split(self):
        return [1]

    @property
    def nparams(self):
        return len(self.psi_fns) + 1
