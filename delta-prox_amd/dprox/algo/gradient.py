"""Forward-backward splitting for  f(x) + g(x)  with f smooth and g proxable (reference dprox/algo/pgd.py:8-54):

    x_{k+1} = prox_{lam_k g}( x_k - rho_k * grad f(x_k) )

``grad f`` of a ``sum_squares`` term is two passes of hand-written kernels (K x - b, then K^T) and the forward step is one
fused AXPY with a per-image step size (``dpx_lincomb``); the backward step is the proxable term's own HIP prox.
"""
import torch
from typing import List, Sequence

from .. import _ops as ops
from ..proxfn import ProxFn
from .driver import Algorithm


def _is_smooth(fn: ProxFn) -> bool:
    """a term takes the gradient role when it exposes ``grad(x)`` (``sum_squares``: proxfn/sum_square.py:29-32)"""
    return callable(getattr(fn, "grad", None))


def forward_step(x, step, gradient):
    """x - step * gradient with ``step`` a 0-d tensor (shared) or a [B] tensor (per image)"""
    coef = -step if getattr(step, "ndim", 0) else -float(step)
    return ops.lincomb([(1.0, x), (coef, gradient)])


class ProximalGradientDescent(Algorithm):
    """state = [x]; exactly two terms, at least one of them smooth (the reference's restriction and error messages)"""

    @classmethod
    def partition(cls, prox_fns: List[ProxFn]):
        terms: Sequence[ProxFn] = list(prox_fns)
        if len(terms) != 2:
            raise ValueError("Proximal gradient descent only supports two proximal functions for now.")
        smooth = [fn for fn in terms if _is_smooth(fn)]
        if not smooth:
            raise ValueError("Proximal gradient descent requires at least one proximal function is differentiable.")
        proxable = [fn for fn in terms if all(fn is not s for s in smooth)]
        return proxable, smooth

    def __init__(self, psi_fns, omega_fns, *unused_args, **unused_kwargs):
        super().__init__(psi_fns, omega_fns)
        self.diff_fn, self.prox_fn = omega_fns[0], psi_fns[0]      # names kept: user code inspects them on the reference

    # ---- Algorithm protocol ---------------------------------------------------------------------------------
    def initialize(self, x0):
        return [x0]

    def iters(self, state, rhos, lams, max_iter, pbar=False, callback=None):
        plan = None
        if callback is None and not pbar and max_iter >= 1:
            self.Kall.update_vars([state[0]])           # (the terms' constant offsets are evaluated around the variable's value)
            plan = self._fused_plan(state[0], rhos, lams)
        if plan is None:
            return super().iters(state, rhos, lams, max_iter, pbar, callback=callback)
        # the whole solve as ONE call: the iterate's spectrum stays resident, two kernels per iteration (dpx_pgd_run)
        from .fused import schedule_table
        kind, gram, ktb = plan
        x = state[0].clone()
        B = int(x.shape[0])
        rho_tab = schedule_table(rhos, max_iter, B, x.device)
        lam_tab = schedule_table(lams[self.prox_fn], max_iter, B, x.device)
        self._notify_all_op_current_step(max_iter - 1)
        from . import fused
        from .. import _backend as be
        C, H, W = (int(d) for d in x.shape[1:])
        chains = fused.sub_batch_chains(B, C, H, W) if (x.is_cuda and (ktb is None or ktb.shape == x.shape)) else 1
        streams = fused.chain_streams(x.device, chains) if chains > 1 else None
        if streams is None:
            ops.pgd_run(x, ktb, gram, kind, float(self.prox_fn.alpha), rho_tab, lam_tab, max_iter)
        else:
            # independent sub-batch chains on separate streams (fused.FusedADMM._run_chains: one chain's column pass beside the other's row pass)
            main, side = streams
            L = be.lib()
            L.call("dpx_admm_iter_share", chains)
            try:
                bounds = [fused.chain_bounds(B, chains, c) for c in range(chains)]
                # every chain's workspace lives until the caller's stream has waited for all chains (a block handed back earlier could be
                # given to the next chain while the previous one still runs)
                wss = [ops._bytes(L.query("dpx_spectrum_bytes", (b1 - b0) * C, H, W), x.device) for b0, b1 in bounds]
                tabs = [(fused._chain_table(rho_tab, b0, b1), fused._chain_table(lam_tab, b0, b1)) for b0, b1 in bounds]
                for st in side:
                    st.wait_stream(main)
                for (b0, b1), ws, (rt, lt), st in zip(bounds, wss, tabs, [main] + side):
                    with torch.cuda.stream(st):
                        ops.pgd_run(x[b0:b1], None if ktb is None else ktb[b0:b1], gram, kind, float(self.prox_fn.alpha), rt, lt, max_iter, ws=ws)
                for st in side:
                    main.wait_stream(st)
                del wss
            finally:
                L.call("dpx_admm_iter_share", 1)
        self.Kall.update_vars([x])
        return [x]

    def _fused_plan(self, x, rhos, lams):
        """(prox code, |OTF|^2 table, K^T b) when the iteration is a circular-convolution least-squares term plus a closed-form
        prox of the variable itself on a power-of-two plane, and nothing wants gradients; None = op by op"""
        import torch
        from ..proxfn.quadratic import sum_squares
        from .fused import _psi_linop_code, _psi_prox_code
        from .. import _backend as be
        f, g = self.diff_fn, self.prox_fn
        if type(f) is not sum_squares or f.alpha != 1 or f.beta != 1 or g not in lams:
            return None
        kind = _psi_prox_code(g)
        if kind not in (be.PROX_NORM1, be.PROX_NONNEG, be.PROX_SUMSQ) or _psi_linop_code(g.linop) != be.LIN_IDENTITY or g.offset is not None:
            return None
        if x.ndim != 4 or x.dtype != torch.float32 or len(self.Kall.variables) != 1:
            return None
        if torch.is_grad_enabled() and any(t.requires_grad for t in [x, rhos] + list(lams.values())):
            return None
        if not ops.pgd_supported(x.shape[2], x.shape[3], kind):
            return None
        tables = f.gram_tables(x)
        return None if tables is None else (kind, tables[0], tables[1])

    def _iter(self, state, rho, lam):
        (x,) = state
        parts = getattr(self.diff_fn, "grad_parts", lambda t: None)(x)
        if parts is not None:                       # x - rho (K^T K x - K^T b) as one fused AXPY
            gram_x, ktb = parts
            coef = rho if getattr(rho, "ndim", 0) else float(rho)
            terms = [(1.0, x), (-coef, gram_x)] + ([] if ktb is None else [(coef, ktb)])
            moved = ops.lincomb(terms)
        else:
            moved = forward_step(x, rho, self.diff_fn.grad(x))
        return [self.prox_fn.prox(moved, lam[self.prox_fn])]

    @property
    def nparams(self):
        return 1 + len(self.psi_fns)

    @property
    def state_split(self):
        return [1]
