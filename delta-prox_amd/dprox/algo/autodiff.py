"""Reverse-mode differentiation of the fused ADMM iteration (config 5: unrolled training).

The reference differentiates its solvers with plain PyTorch autograd through every eager op of the iteration
(dprox/algo/admm.py:49-59 under specialization/unroll.py:14-58; README.md:93-116 trains the rho / lambda schedules).
Here the iteration is three hand-written stages; each gets a hand-written backward built from the same HIP primitives:

    rhs = rho * sum_i K_i^T (v_i - u_i)           linear:      g_v_i = rho K_i g,  g_u_i = -g_v_i,  g_rho = <g, rhs> / rho
    x   = S_rho(rhs + sum_Omega K^T b)            self-adjoint Fourier solve M = F^-1 diag(1/den) F:
                                                  g_rhs = M g,  g_b = K g_rhs,  g_rho = -<g_rhs, (sum_Psi K_i^T K_i) x>
    d_i = K_i x + u_i ; v_i = prox(d_i, lam_i) ;  u_i' = d_i - v_i
                                                  g_d = J_prox^T (g_v - g_u') + g_u',  g_x += K_i^T g_d,  g_u = g_d,
                                                  g_lam = alpha <g_v - g_u', d prox / d lam>

(K_i in {I, grad_H, grad_W}; prox in {soft-threshold, nonneg, v/(1+2 lam)}.)  torch.autograd only chains these
Functions and carries the [B]-sized schedule entries; every image-sized pass runs in libdpx_hip.so.
"""
import os

import torch

from .. import _backend as be
from .. import _ops as ops
from ..linop.fourier import _FullOtf

_DIM = {be.LIN_GRAD_H: 0, be.LIN_GRAD_W: 1}


def _K(code, x):
    return x if code == be.LIN_IDENTITY else ops.grad(x, _DIM[code], adjoint=False)


def _KT(code, y):
    return y if code == be.LIN_IDENTITY else ops.grad(y, _DIM[code], adjoint=True)


def _scaled(coef_b, x, sign=1.0):
    """sign * coef[b] * x[b]"""
    return ops.lincomb([(coef_b if sign == 1.0 else coef_b * sign, x)])


class _LinApply(torch.autograd.Function):
    """y = K x for K in {I, grad_H, grad_W} (the initial split variables v_i = K_i x0, admm.py:61-67)"""

    @staticmethod
    def forward(ctx, code, x):
        ctx.code = code
        return _K(code, x.contiguous()).clone() if code == be.LIN_IDENTITY else _K(code, x.contiguous())

    @staticmethod
    def backward(ctx, g):
        return None, _KT(ctx.code, g.contiguous())


class _LinComb(torch.autograd.Function):
    """sum_i c_i x_i with python-float coefficients"""

    @staticmethod
    def forward(ctx, coefs, *xs):
        ctx.coefs = coefs
        return ops.lincomb([(float(c), x.contiguous()) for c, x in zip(coefs, xs)])

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        return (None, *[g if c == 1.0 else ops.lincomb([(float(c), g)]) for c in ctx.coefs])


class _Rhs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, codes, rho, *vu):
        n = len(codes)
        v, u = vu[:n], vu[n:]
        rhs = torch.empty_like(v[0])
        specs = [dict(linop=lc, prox=pc if pc != be.PROX_EXTERNAL else be.PROX_NORM1, alpha=1.0, lam=None, v=v[i].contiguous(),
                      u=u[i].contiguous()) for i, (lc, pc) in enumerate(codes)]
        ctx.keep = specs
        ops.admm_rhs(rhs, None, rho.contiguous(), ops.make_terms(specs), n)
        ctx.codes = codes
        ctx.save_for_backward(rho, rhs)
        return rhs

    @staticmethod
    def backward(ctx, g):
        rho, rhs = ctx.saved_tensors
        gv, gu, g_rho = ops.admm_rhs_bwd(g.contiguous(), rhs, rho, [lc for lc, _ in ctx.codes])
        return (None, g_rho, *gv, *gu)


class _Solve(torch.autograd.Function):
    """inputs behind (plan, rhs, rho): the full OTFs of plan.doe (conv_doe terms whose PSF is being trained), then the offsets"""

    @staticmethod
    def forward(ctx, plan, rhs, rho, *rest):
        t0, c0, t1, c1 = plan.diag
        x = ops.fourier_solve(rhs.contiguous(), t0, t1, c0, c1, rho, plan.eps, spec_add=plan.FK)
        ctx.plan = plan
        nd = len(plan.doe)
        ctx.save_for_backward(x, rho, *[o.detach() for o in rest[:nd]])
        return x

    @staticmethod
    def backward(ctx, gx):
        plan = ctx.plan
        x, rho, *otfs = ctx.saved_tensors
        nd = len(plan.doe)
        t0, c0, t1, c1 = plan.diag
        g_rhs = ops.fourier_apply_inv(gx.contiguous(), t0, t1, c0, c1, rho, plan.eps)
        # d x / d rho = -M (sum_Psi K_i^T K_i) x : one stencil + reduction pass
        g_rho = ops.admm_solve_rho_grad(g_rhs, x, [lc for lc, _ in plan.codes])
        g_otfs = [None] * nd
        if any(ctx.needs_input_grad[3:3 + nd]):
            # dL/dO = (1/HW) sum_b [conj(A) Y - 2 Re(A conj X) O],  A = fft2(g_rhs), X = fft2(x)   (dpx_otf_grad)
            A = ops.cfft2(g_rhs, inverse=False, centred=False, ortho=False)
            X = ops.cfft2(x, inverse=False, centred=False, ortho=False)
            for j, (need, (_, Yhat)) in enumerate(zip(ctx.needs_input_grad[3:3 + nd], plan.doe)):
                if need:
                    g_otfs[j] = ops.otf_grad(A, X, Yhat, otfs[j])
        g_offs = []
        for need, otf in zip(ctx.needs_input_grad[3 + nd:], plan.omega_otfs):
            if not need:
                g_offs.append(None)
            else:
                g_offs.append(g_rhs if otf is None else ops.fft_conv(g_rhs, otf, conj=False))
        return (None, g_rhs, g_rho, *g_otfs, *g_offs)


class _ZUpdate(torch.autograd.Function):
    """the closed-form prox terms `idx` of the plan (deep priors are chained separately, see run())"""

    @staticmethod
    def forward(ctx, plan, idx, x, *lam_u):
        n = len(idx)
        lams, us = lam_u[:n], lam_u[n:]
        x = x.contiguous()
        vs = [torch.empty_like(x) for _ in idx]
        uos = [torch.empty_like(x) for _ in idx]
        specs = [dict(linop=plan.codes[i][0], prox=plan.codes[i][1], alpha=float(plan.psi[i].alpha), lam=lam.contiguous(), v=v,
                      u=u.contiguous(), u_out=uo) for i, lam, u, v, uo in zip(idx, lams, us, vs, uos)]
        ctx.keep = specs
        ops.admm_zupdate(x, ops.make_terms(specs), n)
        ctx.plan, ctx.idx = plan, idx
        ctx.save_for_backward(*lams, *vs)
        return (*vs, *uos)

    @staticmethod
    def backward(ctx, *g):
        plan = ctx.plan
        n = len(ctx.idx)
        saved = ctx.saved_tensors
        lams, vs = saved[:n], saved[n:]
        gvs, guos = g[:n], g[n:]
        specs = [dict(linop=plan.codes[k][0], prox=plan.codes[k][1], alpha=float(plan.psi[k].alpha), lam=lams[i], v=vs[i], gv=gvs[i],
                      gu_new=guos[i]) for i, k in enumerate(ctx.idx)]
        gx, gus, glams = ops.admm_zupdate_bwd(specs, tuple(vs[0].shape), vs[0].device)
        return (None, None, gx, *glams, *gus)


def needs_grad(x0, rhos, lams, offsets, state_tensors=()):
    if not torch.is_grad_enabled():
        return False
    ts = [x0, rhos] + list(lams.values()) + [o for o in offsets if o is not None] + list(state_tensors)
    return any(isinstance(t, torch.Tensor) and t.requires_grad for t in ts)


class DiffPlan:
    """what the three Functions need from a recognised problem (built by FusedADMM.run_differentiable)"""

    def __init__(self, codes, psi, diag, FK, omega_otfs, eps, hist_bf16=False, doe=()):
        self.codes, self.psi, self.diag, self.FK, self.omega_otfs, self.eps = codes, psi, diag, FK, omega_otfs, eps
        self.hist_bf16 = hist_bf16        # keep the backward pass's history (rhs, x, v_i per iteration) in bf16
        # conv_doe data terms whose PSF requires grad: [(full OTF as an autograd tensor (_FullOtf), fft2 of the term's offset or None)]
        self.doe = list(doe)


class _SchedRows(torch.autograd.Function):
    """a [T'] schedule (T' >= T) as a contiguous [T, B] table: one copy kernel forward, one reduction backward (the chain of slice /
    reshape / expand / contiguous nodes it replaces runs three kernels per table in the backward pass of a training step)"""

    @staticmethod
    def forward(ctx, v, T, B):
        ctx.n = int(v.shape[0])
        return v[:T].reshape(T, 1).expand(T, B).contiguous()

    @staticmethod
    def backward(ctx, g):
        gs = g.sum(dim=1)
        if ctx.n != gs.shape[0]:
            gs = torch.cat([gs, gs.new_zeros(ctx.n - gs.shape[0])])
        return gs, None, None


def _sched_table(vals, T, B, dev):
    """0-d / [T] / [B,T] schedule -> [T,B] device tensor, differentiably, in ONE set of tiny torch ops per solve (row `it`
    is then a contiguous view: no per-iteration kernels)"""
    v = vals if isinstance(vals, torch.Tensor) else torch.as_tensor(vals, dtype=torch.float32)
    v = v.to(device=dev, dtype=torch.float32)
    if v.ndim == 0:
        return v.reshape(1, 1).expand(T, B).contiguous()
    if v.ndim == 1:
        return _SchedRows.apply(v, T, B) if v.requires_grad else v[:T].reshape(T, 1).expand(T, B).contiguous()
    return v[:, :T].t().expand(T, B).contiguous()


class _UnrolledClosed(torch.autograd.Function):
    """All `T` iterations of a problem whose Psi terms are closed-form proxes as ONE autograd node whose forward and backward
    are single C calls (``dpx_admm_unrolled_forward`` / ``_backward``: the same three forward stages and three backward
    kernels per iteration as _Rhs / _Solve / _ZUpdate, sequenced on the C side).  Config 5 is issue-bound from Python: ~170
    launches, 2.3 ms of kernels in a 4.2 ms step.  Inputs: rho_tab [T,B], lam_tabs[i] [T,B], v_i, u_i, offsets; outputs:
    x, v_i, u_i after T iterations (views of the saved history buffer)."""

    @staticmethod
    def _common(plan, dev, shape):
        import ctypes
        n = len(plan.codes)
        B, C, H, W = shape
        lin = (ctypes.c_int * n)(*[lc for lc, _ in plan.codes])
        prx = (ctypes.c_int * n)(*[pc for _, pc in plan.codes])
        alp = (ctypes.c_float * n)(*[float(fn.alpha) for fn in plan.psi])
        t0, c0, t1, c1 = plan.diag
        dd = ops.denominator(t0, c0, t1, c1, C, H, W, dev)
        return lin, prx, alp, dd, ops.fft_table(H, W, dev), ops.spectrum_ws(B * C, H, W, dev)

    @staticmethod
    def forward(ctx, plan, T, rho_tab, *rest):
        import ctypes
        n = len(plan.codes)
        lam_tabs, v, u, offs = rest[:n], rest[n:2 * n], rest[2 * n:3 * n], rest[3 * n:]
        ctx.set_materialize_grads(False)       # outputs the loss does not use (v_i, u_i) arrive as None in backward, not as zero planes
        v = [t.contiguous() for t in v]
        u = [t.contiguous() for t in u]
        shape, dev = tuple(v[0].shape), v[0].device
        B, C, H, W = shape
        # schedules handed over as they are -- [T'] vectors (run() does that when every one of them is): their [T, B] tables are ONE copy
        # kernel here and their gradients two reductions in backward (a table per schedule through _SchedRows: three copies, three reductions
        # and two more copies when the strided rows become .grad -- in a 1-ms training step)
        ctx.sched_len = None
        if rho_tab.ndim == 1:
            ctx.sched_len = [int(t.shape[0]) for t in (rho_tab, *lam_tabs)]
            tabs = torch.cat([t[:T].reshape(1, T, 1).expand(1, T, B) for t in (rho_tab, *lam_tabs)])
            rho_tab, lam_tabs = tabs[0], [tabs[1 + i] for i in range(n)]
        rho_tab = rho_tab.contiguous()
        lam_tabs = [t.contiguous() for t in lam_tabs]
        lin, prx, alp, dd, table, sws = _UnrolledClosed._common(plan, dev, shape)
        # the state is ADMM.initialize(x0) untouched and was never computed (fused.lazy_initial_state): the C side forms the first right-hand
        # side from x0 itself (two stencil passes, two zero fills and their copies less per training step)
        fresh_x0 = getattr(plan, "fresh_x0", None)
        plan.fresh_x0 = None
        vp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in v])
        up = (ctypes.c_void_p * n)(*[t.data_ptr() for t in u])
        lp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in lam_tabs])
        if plan.hist_bf16:
            L = be.lib()
            hist = torch.empty((T, 2 + n) + shape, dtype=torch.bfloat16, device=dev)
            work = ops.workspace("unrolled_work", L.query("dpx_admm_unrolled_work_bytes_bf16", n, B, C, H, W), dev)
            x_out = torch.empty(shape, dtype=torch.float32, device=dev)
            v_out = [torch.empty(shape, dtype=torch.float32, device=dev) for _ in range(n)]
            u_out = [torch.empty(shape, dtype=torch.float32, device=dev) for _ in range(n)]
            L.call("dpx_admm_unrolled_forward_bf16", be.ptr(hist), be.ptr(work), be.ptr(x_out), (ctypes.c_void_p * n)(*[t.data_ptr() for t in v_out]),
                   (ctypes.c_void_p * n)(*[t.data_ptr() for t in u_out]), vp, up, lin, prx, alp, n, be.ptr(rho_tab), lp, T, be.ptr(plan.FK), be.ptr(dd),
                   ctypes.c_float(plan.eps), B, C, H, W, be.ptr(table), be.ptr(sws), be.ptr(fresh_x0), be.stream())
            ctx.plan, ctx.T, ctx.n_off, ctx.shape = plan, T, len(offs), shape
            ctx.save_for_backward(rho_tab, *lam_tabs)
            ctx.hist = hist
            return (x_out, *v_out, *u_out)
        hist = torch.empty((T, 2 + 2 * n) + shape, dtype=torch.float32, device=dev)
        be.lib().call("dpx_admm_unrolled_forward", be.ptr(hist), vp, up, lin, prx, alp, n, be.ptr(rho_tab), lp, T, be.ptr(plan.FK), be.ptr(dd),
                      ctypes.c_float(plan.eps), B, C, H, W, be.ptr(table), be.ptr(sws), be.ptr(fresh_x0), be.stream())
        ctx.plan, ctx.T, ctx.n_off, ctx.shape = plan, T, len(offs), shape
        ctx.save_for_backward(rho_tab, *lam_tabs)
        ctx.hist = hist
        last = hist[T - 1]
        return (last[1], *[last[2 + i] for i in range(n)], *[last[2 + n + i] for i in range(n)])

    @staticmethod
    def backward(ctx, gx, *gvu):
        import ctypes
        plan, T, hist, shape = ctx.plan, ctx.T, ctx.hist, ctx.shape
        n = len(plan.codes)
        B, C, H, W = shape
        dev = hist.device
        rho_tab, *lam_tabs = ctx.saved_tensors
        lin, prx, alp, dd, table, sws = _UnrolledClosed._common(plan, dev, shape)
        keep = [None if g is None else g.contiguous() for g in (gx, *gvu)]
        gxp = keep[0]
        gvi = (ctypes.c_void_p * n)(*[None if g is None else g.data_ptr() for g in keep[1:1 + n]])
        gui = (ctypes.c_void_p * n)(*[None if g is None else g.data_ptr() for g in keep[1 + n:1 + 2 * n]])
        gv0 = [torch.empty(shape, dtype=torch.float32, device=dev) for _ in range(n)]
        gu0 = [torch.empty(shape, dtype=torch.float32, device=dev) for _ in range(n)]
        g_rho = torch.empty(T, B, dtype=torch.float32, device=dev)
        g_lam = torch.empty(T, n, B, dtype=torch.float32, device=dev)
        need_off = ctx.needs_input_grad[3 + 3 * n:]
        g_off = [torch.empty(shape, dtype=torch.float32, device=dev) if need else None for need in need_off]
        n_off = ctx.n_off
        gop = (ctypes.c_void_p * max(n_off, 1))(*[None if t is None else t.data_ptr() for t in g_off])
        otf = (ctypes.c_void_p * max(n_off, 1))(*[None if o is None else o.data_ptr() for o in plan.omega_otfs[:n_off]])
        lp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in lam_tabs])
        L = be.lib()
        bf16 = hist.dtype == torch.bfloat16
        ws = ops.workspace("unrolled_bwd", L.query("dpx_admm_unrolled_bwd_ws_bytes_bf16" if bf16 else "dpx_admm_unrolled_bwd_ws_bytes", n, B, C, H, W), dev)
        L.call("dpx_admm_unrolled_backward_bf16" if bf16 else "dpx_admm_unrolled_backward", be.ptr(hist), be.ptr(gxp), gvi, gui, (ctypes.c_void_p * n)(*[t.data_ptr() for t in gv0]),
               (ctypes.c_void_p * n)(*[t.data_ptr() for t in gu0]), be.ptr(g_rho), be.ptr(g_lam), gop, otf, n_off, lin, prx, alp, n,
               be.ptr(rho_tab), lp, T, be.ptr(dd), ctypes.c_float(plan.eps), B, C, H, W, be.ptr(table), be.ptr(sws), be.ptr(ws), be.stream())
        if ctx.sched_len is not None:                     # [T'] schedules: sum over the images (two reductions; the terms' rows come out contiguous:
            g_r, g_l = g_rho.sum(dim=1), g_lam.permute(1, 0, 2).sum(dim=2)        # no copy when they become .grad), zero beyond the T iterations run
            pad = lambda g, m: g if m == T else torch.cat([g, g.new_zeros(m - T)])
            return (None, None, pad(g_r, ctx.sched_len[0]), *[pad(g_l[i], ctx.sched_len[1 + i]) for i in range(n)], *gv0, *gu0, *g_off)
        return (None, None, g_rho, *[g_lam[:, i] for i in range(n)], *gv0, *gu0, *g_off)


def run(plan: DiffPlan, state, rhos, lams, max_iter, diff_offsets):
    """max_iter differentiable ADMM iterations from `state` = (x, [v_i], [u_i]); returns the new state.
    When x requires grad but the split variables are not connected to it (the state came straight from
    ``initialize``), v_i = K_i x is rebuilt differentiably."""
    x, v, u = state
    B, dev = int(x.shape[0]), x.device
    n = len(plan.codes)
    codes = tuple(plan.codes)
    v, u = list(v), list(u)
    if x.requires_grad:
        v = [_LinApply.apply(lc, x) if (t.grad_fn is None and not t.requires_grad) else t for (lc, _), t in zip(codes, v)]
    closed = tuple(i for i, (_, pc) in enumerate(codes) if pc != be.PROX_EXTERNAL)
    ext = [i for i, (_, pc) in enumerate(codes) if pc == be.PROX_EXTERNAL]
    doe_otfs = [o for o, _ in plan.doe]
    one_node = not ext and n > 0 and max_iter > 0 and not doe_otfs and not os.environ.get("DPX_UNROLL_CHAIN")
    if one_node:
        # [T'] schedules on the device go to the node as they are (its own single copy / single reduction, see _UnrolledClosed.forward)
        sched = [rhos] + [lams[fn] for fn in plan.psi]
        if all(isinstance(t, torch.Tensor) and t.ndim == 1 and t.shape[0] >= max_iter and t.device == dev and t.dtype == torch.float32 for t in sched):
            out = _UnrolledClosed.apply(plan, max_iter, *sched, *v, *u, *diff_offsets)
            return out[0], list(out[1:1 + n]), list(out[1 + n:1 + 2 * n])
    rho_tab = _sched_table(rhos, max_iter, B, dev)
    lam_tabs = [_sched_table(lams[fn], max_iter, B, dev) for fn in plan.psi]
    if one_node:
        out = _UnrolledClosed.apply(plan, max_iter, rho_tab, *lam_tabs, *v, *u, *diff_offsets)
        return out[0], list(out[1:1 + n]), list(out[1 + n:1 + 2 * n])
    for it in range(max_iter):
        rho = rho_tab[it]
        lam = [lt[it] for lt in lam_tabs]
        rhs = _Rhs.apply(codes, rho, *v, *u)
        x = _Solve.apply(plan, rhs, rho, *doe_otfs, *diff_offsets)
        nv, nu = list(v), list(u)
        if closed:
            out = _ZUpdate.apply(plan, closed, x, *[lam[i] for i in closed], *[u[i] for i in closed])
            for k, i in enumerate(closed):
                nv[i], nu[i] = out[k], out[len(closed) + k]
        for i in ext:                                   # deep prior on the identity: v = D(x + u, sigma), u' = x + u - v
            fn = plan.psi[i]
            d = _LinComb.apply((1.0, 1.0), x, u[i])
            fn.step = it
            nv[i] = fn._prox(d, lam[i] * float(fn.alpha))
            nu[i] = _LinComb.apply((1.0, -1.0), d, nv[i])
        v, u = nv, nu
    return x, v, u
