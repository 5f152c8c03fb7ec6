"""Operator-splitting solvers: ``ADMM``, ``LinearizedADMM``, ``ADMM_vxu``, ``HQS``, ``PockChambolle``
(reference dprox/algo/admm.py:25-120, hqs.py:4-20, pc.py:6-40, invert.py:5-14).

Variable splitting follows the reference: exact-type ``sum_squares`` terms (and the first
``ext_sum_squares``) form Omega and share x; every other term gets its own split variable.
``ADMM`` problems whose graph is recognised by ``fused.plan_admm`` run the fused HIP iteration
(3 kernel stages per iteration); everything else, including user plugins, runs op-by-op on the
same HIP primitives through ``_iter``.
"""
from typing import List

import torch

from .. import _ops as ops
from ..linalg import LinearSolveConfig
from ..linop import Variable, adjoint, eval
from ..proxfn import ProxFn, ext_sum_squares, least_squares, sum_squares
from .driver import Algorithm, expand
from . import fused


def get_least_square_solver(psi_fns, omega_fns, try_diagonalize, try_freq_diagonalize, linear_solve_config):
    prox_fns = list(psi_fns) + list(omega_fns)
    ext_sq = [fn for fn in omega_fns if isinstance(fn, ext_sum_squares)]
    for fn in ext_sq:
        other = [f for f in prox_fns if f is not fn]
        if all(isinstance(f.linop, Variable) for f in other):
            return ext_sq[0].setup([f.b for f in omega_fns if f is not fn and f not in ext_sq])
    return least_squares(omega_fns, psi_fns, try_diagonalize, try_freq_diagonalize, linear_solve_config=linear_solve_config)


def _add(*terms):
    return ops.lincomb([(c, t.contiguous()) for c, t in terms])


class ADMM(Algorithm):
    @classmethod
    def partition(cls, prox_fns: List[ProxFn]):
        omega_fns, taken = [], False
        for fn in prox_fns:
            if not taken and isinstance(fn, ext_sum_squares):
                omega_fns.append(fn)
                taken = True
            elif type(fn) == sum_squares:
                omega_fns.append(fn)
        psi_fns = [fn for fn in prox_fns if not any(fn is o for o in omega_fns)]
        return psi_fns, omega_fns

    def __init__(self, psi_fns, omega_fns, try_diagonalize=True, try_freq_diagonalize=True,
                 linear_solve_config=LinearSolveConfig()):
        super().__init__(psi_fns, omega_fns)
        self.least_square = get_least_square_solver(psi_fns, omega_fns, try_diagonalize, try_freq_diagonalize,
                                                    linear_solve_config)
        self.use_fused = True
        self.last_path = None           # "fused" | "generic": which engine ran the last solve

    def initialize(self, x0, v=None):
        x = x0
        self.Kall.update_vars([x])
        derived = v is None
        if v is None:
            v = self.K.forward(x, return_list=True)
            if v is None:
                v = []
        v = [e if e is not x else e.clone() for e in v]
        u = [torch.zeros_like(e) for e in v]
        # the fused path recognises this state as long as nobody has written to it (fused.fresh_state): v_i = K_i x0, u_i = 0 exactly
        self._fresh = (x, v, u, [t._version for t in [x] + v + u]) if derived and all(isinstance(t, torch.Tensor) for t in [x] + v) else None
        return x, v, u

    def _plan_for(self, x):
        """the fused plan of this problem for an iterate like ``x`` (the pattern match depends on the problem graph and on the
        iterate's shape / dtype: done once per shape and per configuration of the terms, fused.plan_fingerprint)"""
        key = (tuple(x.shape), x.dtype, fused.plan_fingerprint(self)) if isinstance(x, torch.Tensor) else None
        hit = getattr(self, "_plan_cache", None)
        if key is not None and hit is not None and hit[0] == key:
            return hit[1]
        plan = fused.plan_admm(self, (x, [], []))
        self._plan_cache = (key, plan)
        return plan

    def _initial_state(self, x0, **kwargs):
        """solve()'s initial state: for problems the two-kernel iteration takes, the split variables are allocated but not computed
        (fused.lazy_initial_state: v_i = K_i x0, u_i = 0 are implied and never read); ``initialize`` itself stays eager"""
        if type(self) is ADMM and self.use_fused and not kwargs:
            st = fused.lazy_initial_state(self, x0, self._plan_for(x0))
            if st is not None:
                return st
        return self.initialize(x0, **kwargs)

    def iters(self, state, rhos, lams, max_iter, pbar=False, callback=None):
        plan = self._plan_for(state[0]) if (self.use_fused and type(self) is ADMM) else None
        if plan is not None:
            self.last_path = "fused"
            return plan.run(state, rhos, lams, max_iter, pbar, callback)
        fused.materialize_state(self, state)                   # (every other path reads the split variables)
        if self.use_fused and type(self) in (ADMM, LinearizedADMM):
            plan = fused.plan_split_cg(self, state, rhos, lams)     # CG x-update, every Psi term on x itself (config 4)
            if plan is not None:
                self.last_path = "fused-cg"
                return plan.run(state, rhos, lams, max_iter, pbar, callback)
        if type(self) is LinearizedADMM:
            plan = _fused_closed_form(self, state, rhos, lams)      # Fourier x-update, Psi terms on x / grad x: nested-stencil rhs pass
            if plan is not None:
                self.last_path = "fused"
                return plan.run_stencil(state, rhos, lams, max_iter, "ladmm", pbar, callback)
        self.last_path = "generic"
        self._fresh = None
        return super().iters(state, rhos, lams, max_iter, pbar, callback)

    def _prox_dual(self, Kx, v, u, lam):
        for i, fn in enumerate(self.psi_fns):
            t = _add((1.0, Kx[i]), (1.0, u[i]))
            v[i] = fn.prox(t, lam=lam[fn])
            u[i] = _add((1.0, t), (-1.0, v[i]))

    def _iter(self, state, rho, lam):
        x, v, u = state
        b = [_add((1.0, v[i]), (-1.0, u[i])) for i in range(len(self.psi_fns))]
        x = self.least_square.solve(b, rho)
        Kx = self.K.forward(x, return_list=True)
        self._prox_dual(Kx, v, u, lam)
        return x, v, u

    @property
    def nparams(self):
        return len(self.psi_fns) + 1

    @property
    def state_split(self):
        return [1, [len(self.psi_fns)], [len(self.psi_fns)]]


class LinearizedADMM(ADMM):
    def _iter(self, state, rho, lam):
        x, v, u = state
        b = []
        for i, fn in enumerate(self.psi_fns):
            tmp = _add((1.0, eval(fn.linop, x)), (-1.0, v[i]), (1.0, u[i]))
            tmp = adjoint(fn.linop, tmp)
            b.append(_add((1.0, x), (-1.0, tmp)))          # the reference fixes the step coefficient to 1 (admm.py:87)
        x = self.least_square.solve(b, rho)
        Kx = self.K.forward(x, return_list=True)
        self._prox_dual(Kx, v, u, lam)
        return x, v, u


def _fused_closed_form(solver, state, rhos, lams):
    """the fused ADMM's plan when every Psi prox is closed-form and no gradient is requested, else None"""
    plan = fused.plan_admm(solver, state) if solver.use_fused else None
    if plan is None or len(state[1]) == 0 or any(pc == fused.be.PROX_EXTERNAL for _, pc in plan.codes):
        return None
    tensors = [state[0], rhos] + list(lams.values()) + [t for part in state[1:] for t in (part if isinstance(part, (list, tuple)) else [part])]
    if torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors):
        return None
    return plan


class ADMM_vxu(ADMM):
    """update order v, x, u"""

    def iters(self, state, rhos, lams, max_iter, pbar=False, callback=None):
        plan = _fused_closed_form(self, state, rhos, lams)
        if plan is not None:
            self.last_path = "fused"
            return plan.run(state, rhos, lams, max_iter, pbar, callback, vxu=True)
        self.last_path = "generic"
        return Algorithm.iters(self, state, rhos, lams, max_iter, pbar, callback)

    def _iter(self, state, rho, lam):
        z, x, u = state
        Kz = self.K.forward(z, return_list=True)
        for i, fn in enumerate(self.psi_fns):
            x[i] = fn.prox(_add((1.0, Kz[i]), (-1.0, u[i])), lam=lam[fn])
        b = [_add((1.0, x[i]), (1.0, u[i])) for i in range(len(self.psi_fns))]
        z = self.least_square.solve(b, rho)
        for i, fn in enumerate(self.psi_fns):
            u[i] = _add((1.0, u[i]), (1.0, x[i]), (-1.0, z))
        return z, x, u


class HQS(ADMM):
    def initialize(self, x0):
        x = x0
        self.Kall.update_vars([x])
        z = self.K.forward(x, return_list=True)
        return x, [e if e is not x else e.clone() for e in z]

    def iters(self, state, rhos, lams, max_iter, pbar=False, callback=None):
        """recognised problems (the fused ADMM's criteria, closed-form proxes only, no gradients requested) run the fused
        rhs / Fourier-solve / z stages with the dual variables pinned to zero; everything else op by op"""
        plan = _fused_closed_form(self, state, rhos, lams)
        if plan is not None:
            self.last_path = "fused"
            return plan.run(state, rhos, lams, max_iter, pbar, callback, dual=False)
        self.last_path = "generic"
        return Algorithm.iters(self, state, rhos, lams, max_iter, pbar, callback)

    def _iter(self, state, rho, lam):
        x, z = state
        x = self.least_square.solve(z, rho)
        Kx = self.K.forward(x, return_list=True)
        for i, fn in enumerate(self.psi_fns):
            z[i] = fn.prox(Kx[i].contiguous(), lam=lam[fn])
        return x, z

    @property
    def state_split(self):
        return [1, [len(self.psi_fns)]]


class PockChambolle(ADMM):
    def initialize(self, x0):
        x = x0
        self.Kall.update_vars([x])
        xbar = x.clone()
        z = self.K.forward(x, return_list=True)
        return x, [e if e is not x else e.clone() for e in z], xbar

    def iters(self, state, rhos, lams, max_iter, pbar=False, callback=None):
        plan = _fused_closed_form(self, (state[0], state[1]), rhos, lams) if len(self.omega_fns) > 0 else None
        if plan is not None:
            self.last_path = "fused"
            return plan.run_stencil(state, rhos, lams, max_iter, "pc", pbar, callback)
        self.last_path = "generic"
        return Algorithm.iters(self, state, rhos, lams, max_iter, pbar, callback)

    def _iter(self, state, rho, lam):
        x, z, xbar = state
        Kxbar = self.K.forward(xbar, return_list=True)
        for i, fn in enumerate(self.psi_fns):
            r = lam[fn]
            z[i] = _add((1.0, z[i]), (r, Kxbar[i]))
            z[i] = _add((1.0, z[i]), (-r if isinstance(r, torch.Tensor) else -float(r), fn.prox(z[i], lam=r)))
        Ktz = [adjoint(fn.linop, z[i]) for i, fn in enumerate(self.psi_fns)]
        x_next = [_add((1.0, x), (-1.0, k)) for k in Ktz]
        if len(self.omega_fns) > 0:
            x_next = self.least_square.solve(x_next, rho)
        else:
            x_next = _add(*[(1.0, t) for t in x_next])
        xbar = _add((2.0, x_next), (-1.0, x))
        return x_next, z, xbar

    @property
    def state_split(self):
        return [1, [len(self.psi_fns)], 1]
