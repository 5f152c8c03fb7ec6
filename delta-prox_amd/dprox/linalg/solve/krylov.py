"""Matrix-free conjugate gradients on the HIP primitives (reference dprox/linalg/solve/solver_cg.py:7-136).

Per iteration: one operator application (the caller's kernels), one fused B x B residual Gram pass (its diagonal is
gamma = <r_i, r_i>; the reference's stop rule uses the *spectral* norm of the [B, N] residual matrix and therefore
couples the images of a batch -- solver_cg.py:103-104), one batched dot <p, Ap>, one direction update and one fused
x / r update.  Dots are reduced with wavefront shuffles and are deterministic (no atomics).

The loop is controlled ON THE DEVICE (``ops.CgControl`` -> ``dpx_cg_test / _direction / _update``): the stop test, beta, alpha
and the updates read a small state block in HBM, so the host issues iteration after iteration without waiting for anything;
after convergence the control kernels return at once (the iterate is frozen at the reference's exit point) and the host, which
polls the `done` flag through a pinned buffer without blocking, stops issuing.  ``verbose=True`` and batches of more than 64
systems run the host-paced loop (``_cg_host``), which needs the residual norm on the host.

dtype: float32 is the backend's arithmetic; a float64 right-hand side keeps the whole solve in float64 (``dpx_bdot_f64`` /
``dpx_bgram_f64`` / ``dpx_lincomb_f64`` on the host-paced loop), as the reference's dtype-generic solvers do -- its own tests solve
float64 systems at rtol 1e-8 (tests/linalg/test_linear_solver.py:57-80); other dtypes are computed in float32.
Autograd: ``cg`` called with something to differentiate (``b`` or parameters of an nn.Module ``A``) returns the solution with the
implicit-function backward of ``dprox.linalg.LinearSolve`` (one more solve with ``A`` -- symmetric -- and one operator VJP): the
gradient the reference obtains by back-propagating through its unrolled iterations, at convergence
(tests/linalg/test_linear_solver_torch.py:60-130 compares exactly these two).
``cg2`` / ``pcg`` (solver_cg.py:139-233): the reference's two un-batched variants -- global dots, x0 = ones, absolute stop rules
``<r, r> < rtol`` and ``max|r| < rtol`` -- on the same primitives; their scalars are formed on the host in float64."""
import numpy as np
import torch

from ... import _ops as ops


def bdot(x: torch.Tensor, y: torch.Tensor):
    """batched dot over all non-leading dims -> [B]  (a plain dot for 1-D inputs)"""
    if x.ndim != y.ndim:
        raise ValueError("The input of `bdot` should have the same shape.")
    if x.ndim == 1:
        return ops.bdot(x.reshape(1, -1).contiguous(), y.reshape(1, -1).contiguous())[0]
    return ops.bdot(x.contiguous(), y.contiguous())


def expand(x: torch.Tensor, ref: torch.Tensor):
    while x.ndim < ref.ndim:
        x = x.unsqueeze(-1)
    return x


def ravel(x: torch.Tensor):
    return x if x.ndim == 1 else x.reshape(x.shape[0], -1)


def _as_batch(t):
    return (t.reshape(1, -1), True) if t.ndim == 1 else (t, False)


def _work(b):
    """the right-hand side as the solve's working tensor: detached, contiguous, float64 kept, anything else -> float32"""
    b = b.detach().contiguous()
    return b if b.dtype in (torch.float32, torch.float64) else b.float()


def _wants_grad(A, b):
    if not torch.is_grad_enabled():
        return False
    return b.requires_grad or (isinstance(A, torch.nn.Module) and any(p.requires_grad for p in A.parameters()))


class _ImplicitCG(torch.autograd.Function):
    """x = solver(A, b) with dL/db = A^-1 dL/dx (A symmetric) and dL/dtheta = -(dL/db)^T (dA/dtheta) x"""

    @staticmethod
    def forward(ctx, solver, A, b, kwargs, *theta):
        with torch.no_grad():
            x = solver(A, b.detach(), **kwargs)
        ctx.solver, ctx.A, ctx.kwargs = solver, A, kwargs
        ctx.save_for_backward(x)
        return x

    @staticmethod
    def backward(ctx, gx):
        (x,) = ctx.saved_tensors
        with torch.no_grad():
            gb = ctx.solver(ctx.A, gx.contiguous(), **ctx.kwargs)
        A = ctx.A
        theta = [p for p in A.parameters() if p.requires_grad] if isinstance(A, torch.nn.Module) else []
        gth = ()
        if theta:
            with torch.enable_grad():
                minus_Ax = -A(x.detach())
            gth = torch.autograd.grad((minus_Ax,), theta, grad_outputs=(gb,), allow_unused=True)
        return (None, None, gb, None, *gth)


def cg(A, b, x0=None, rtol=1e-6, max_iters=100, verbose=False, return_iters=False):
    """Solve A x = b for symmetric positive definite A given as a callable."""
    if _wants_grad(A, b) and not return_iters:
        return _differentiable(cg, A, b, dict(x0=x0, rtol=rtol, max_iters=max_iters, verbose=verbose))
    with torch.no_grad():
        return _cg(A, b, x0, rtol, max_iters, verbose, return_iters)


def _differentiable(solver, A, b, kwargs):
    theta = [p for p in A.parameters() if p.requires_grad] if isinstance(A, torch.nn.Module) else []
    return _ImplicitCG.apply(solver, A, b, kwargs, *theta)


def _cg(A, b, x0, rtol, max_iters, verbose, return_iters):
    b = _work(b)
    bb, flat = _as_batch(b)
    B = bb.shape[0]
    if verbose or B > ops.CgControl.MAX_B or b.dtype == torch.float64:
        return _cg_host(A, b, x0, rtol, max_iters, verbose, return_iters)
    apply = (lambda t: A(t.reshape(b.shape)).detach().reshape(bb.shape).float().contiguous())
    if x0 is None:
        x = ops.zeros_like(bb)
        r = ops.lincomb([(1.0, bb)])                      # b - A(0): A is linear, skip the wasted operator application
    else:
        x = x0.detach().reshape(bb.shape).contiguous().float().clone()
        r = ops.lincomb([(1.0, bb), (-1.0, apply(x))])
    p = ops.zeros_like(bb)
    ctl = ops.CgControl(bb, rtol)
    n_it = int(min(max_iters, b.numel()))
    on_gpu = r.is_cuda
    # The host runs at most LAG iterations ahead of what the GPU has confirmed: before issuing iteration `it` it looks at the
    # flags copied out after iteration it - LAG (pinned buffer + event; by then normally complete, so the wait is free) and
    # stops if the solve has converged.  The GPU always has work queued, and at most LAG operator applications are wasted
    # after convergence (the control kernels of those iterations return at once, the iterate is not touched).
    LAG, ring = 2, 4
    pins = [torch.empty(4, dtype=torch.int32, pin_memory=True) for _ in range(ring)] if on_gpu else None
    events, done, n_done = [None] * ring, False, n_it

    def confirmed(k):
        events[k % ring].synchronize()
        return bool(int(pins[k % ring][0])), int(pins[k % ring][1])

    last = -1
    for it in range(n_it):
        if on_gpu and it >= LAG:
            done, nd = confirmed(it - LAG)
            if done:
                n_done = nd
                break
        ctl.test(r)                          # Gram + stop rule + beta on the device
        ctl.direction(p, r)                  # p = r + beta p
        Ap = apply(p)
        ctl.update(x, r, p, Ap)              # <p, Ap>, alpha, x += alpha p, r -= alpha Ap
        last = it
        if on_gpu:
            pins[it % ring].copy_(ctl.flags, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            events[it % ring] = ev
        else:
            fl = ctl.flags.numpy()
            if int(fl[0]):
                done, n_done = True, int(fl[1])
                break
    if on_gpu and not done and return_iters and last >= 0:
        # the exit iteration is wanted on the host: one wait at the end of the solve for the newest flags
        done, nd = confirmed(last)
        if done:
            n_done = nd
    x = x.reshape(b.shape)
    return (x, n_done) if return_iters else x


def _cg_host(A, b, x0=None, rtol=1e-6, max_iters=100, verbose=False, return_iters=False):
    """host-paced variant: the stop test runs on the host (eigvalsh of the Gram matrix read back every iteration)"""
    b = _work(b)
    bb, flat = _as_batch(b)
    B = bb.shape[0]
    apply = (lambda t: A(t.reshape(b.shape)).detach().reshape(bb.shape).to(b.dtype).contiguous())
    if x0 is None:
        x = torch.zeros_like(bb)
        r = bb.clone()                      # b - A(0): A is linear, skip the wasted operator application
    else:
        x = x0.detach().reshape(bb.shape).to(b.dtype).contiguous()
        r = ops.lincomb([(1.0, bb), (-1.0, apply(x))])
    cg_tol = rtol * np.sqrt(np.maximum(ops.bdot(bb, bb).cpu().numpy().astype(np.float64), 0.0))   # rtol * ||b_i||
    n_it = int(min(max_iters, b.numel()))
    p = gamma_1 = None
    done = n_it
    normr = None
    # The stop test needs the residual Gram matrix on the host (spectral norm, eigvalsh).  On the GPU its read-back is
    # asynchronous (pinned buffer + event) and the iteration's updates are enqueued BEFORE the host waits for it, so the
    # device never idles behind the host round trip; if the test then says "converged", the speculative updates are simply
    # dropped (they were written to fresh tensors) -- decisions and results are exactly those of the sequential loop.
    on_gpu = r.is_cuda
    pin = torch.empty((B, B), dtype=b.dtype, pin_memory=True) if on_gpu else None

    def converged(Gh):
        nonlocal normr
        Gh = Gh.astype(np.float64)
        normr = float(np.sqrt(max(np.linalg.eigvalsh((Gh + Gh.T) * 0.5)[-1], 0.0))) if B > 1 else float(np.sqrt(max(Gh[0, 0], 0.0)))
        return bool(np.all(normr <= cg_tol))

    for it in range(n_it):
        G = ops.bgram(r)                                     # [B,B] on device
        if on_gpu:
            pin.copy_(G, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        elif converged(G.cpu().numpy()):
            if verbose:
                print("Converged at CG Iter %03d" % it)
            done = it
            break
        gamma = G.diagonal().contiguous()                    # <r_i, r_i>
        p_new = ops.lincomb([(1.0, r), (gamma / gamma_1, p)]) if it > 0 else r.clone()
        Ap = apply(p_new)
        alpha = gamma / ops.bdot(p_new, Ap)
        x_new = ops.lincomb([(1.0, x), (alpha, p_new)])
        r_new = ops.lincomb([(1.0, r), (-alpha, Ap)])
        if on_gpu:
            ev.synchronize()
            if converged(pin.numpy()):
                if verbose:
                    print("Converged at CG Iter %03d" % it)
                done = it
                break
        x, r, p, gamma_1 = x_new, r_new, p_new, gamma
    else:
        if verbose:
            print(f"Not converged, r norm={normr}")
    x = x.reshape(b.shape)
    return (x, done) if return_iters else x


def _flat1(t):
    return t.reshape(1, -1)


def _dot(x, y):
    """global dot over all elements (the reference's ``x.ravel() @ y.ravel()``) as a python float"""
    return float(ops.bdot(_flat1(x), _flat1(y))[0])


def cg2(A, b, x0=None, rtol=1e-6, max_iters=500, verbose=False):
    """Un-batched conjugate gradients (solver_cg.py:139-170): global dot products over the whole tensor, initial guess ones,
    stop when the squared residual norm falls below ``rtol`` (an ABSOLUTE threshold, as in the reference)."""
    if _wants_grad(A, b):
        return _differentiable(cg2, A, b, dict(x0=x0, rtol=rtol, max_iters=max_iters, verbose=verbose))
    with torch.no_grad():
        b = _work(b)
        apply = lambda t: A(t).detach().to(b.dtype).contiguous()
        x = torch.ones_like(b) if x0 is None else x0.detach().to(b.dtype).contiguous().clone()
        r = ops.lincomb([(1.0, _flat1(b)), (-1.0, _flat1(apply(x)))]).reshape(b.shape)
        d = r.clone()
        rnorm = _dot(r, r)
        for it in range(max_iters):
            Ad = apply(d)
            alpha = rnorm / _dot(d, Ad)
            ops.lincomb([(1.0, _flat1(x)), (alpha, _flat1(d))], out=_flat1(x))
            ops.lincomb([(1.0, _flat1(r)), (-alpha, _flat1(Ad))], out=_flat1(r))
            rnorm2 = _dot(r, r)
            beta, rnorm = rnorm2 / rnorm, rnorm2
            ops.lincomb([(1.0, _flat1(r)), (beta, _flat1(d))], out=_flat1(d))
            if rnorm2 < rtol:
                if verbose:
                    print(f"converge at iter={it}, rtol={rtol}")
                break
        return x


def pcg(A, b, x0=None, rtol=1e-6, max_iters=100, verbose=False, Minv=None):
    """Preconditioned conjugate gradients (solver_cg.py:173-233): ``Minv`` applies the preconditioner (identity if None), initial
    guess ones, global dots, stop when ``max|r| < rtol`` (absolute, as in the reference)."""
    if _wants_grad(A, b):
        return _differentiable(pcg, A, b, dict(x0=x0, rtol=rtol, max_iters=max_iters, verbose=verbose, Minv=Minv))
    with torch.no_grad():
        b = _work(b)
        apply = lambda t: A(t).detach().to(b.dtype).contiguous()
        prec = (lambda t: t) if Minv is None else (lambda t: Minv(t).detach().to(b.dtype).contiguous())
        x = torch.ones_like(b) if x0 is None else x0.detach().to(b.dtype).contiguous().clone()
        r = ops.lincomb([(1.0, _flat1(apply(x))), (-1.0, _flat1(b))]).reshape(b.shape)
        y = prec(r)
        p = ops.lincomb([(-1.0, _flat1(y))]).reshape(b.shape)
        bnorm = float(ops.absmax(b))
        rnorm, it = float("nan"), -1
        for it in range(max_iters):
            Ap = apply(p)
            ry = _dot(r, y)
            alpha = ry / _dot(p, Ap)
            ops.lincomb([(1.0, _flat1(x)), (alpha, _flat1(p))], out=_flat1(x))
            ops.lincomb([(1.0, _flat1(r)), (alpha, _flat1(Ap))], out=_flat1(r))
            y = prec(r)
            beta = _dot(r, y) / ry
            ops.lincomb([(-1.0, _flat1(y)), (beta, _flat1(p))], out=_flat1(p))
            rnorm = float(ops.absmax(r))
            if rnorm < rtol:
                break
        if verbose:
            print(f"#IT: {it + 1}; bnorm: {bnorm:.3e}; rnorm: {rnorm:.3e}; rtol: {rtol:.3e}")
        return x
