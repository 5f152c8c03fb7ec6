"""Matrix-free conjugate gradients on the HIP primitives (reference dprox/linalg/solve/solver_cg.py:7-136).

Per iteration: one operator application (the caller's kernels), one fused B x B residual Gram pass (its diagonal is
gamma = <r_i, r_i>; the reference's stop rule uses the *spectral* norm of the [B, N] residual matrix and therefore
couples the images of a batch -- solver_cg.py:103-104), one batched dot <p, Ap>, one direction update and one fused
x / r update.  Dots are reduced with wavefront shuffles and are deterministic (no atomics).

The loop is controlled ON THE DEVICE (``ops.CgControl`` -> ``dpx_cg_test / _direction / _update``): the stop test, beta, alpha
and the updates read a small state block in HBM, so the host issues iteration after iteration without waiting for anything;
after convergence the control kernels return at once (the iterate is frozen at the reference's exit point) and the host, which
polls the `done` flag through a pinned buffer without blocking, stops issuing.  ``verbose=True`` and batches of more than 64
systems run the host-paced loop (``_cg_host``), which needs the residual norm on the host."""
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


def cg(A, b, x0=None, rtol=1e-6, max_iters=100, verbose=False, return_iters=False):
    """Solve A x = b for symmetric positive definite A given as a callable."""
    b = b.contiguous()
    if b.dtype != torch.float32:
        b = b.float()
    bb, flat = _as_batch(b)
    B = bb.shape[0]
    if verbose or B > ops.CgControl.MAX_B:
        return _cg_host(A, b, x0, rtol, max_iters, verbose, return_iters)
    apply = (lambda t: A(t.reshape(b.shape)).reshape(bb.shape).contiguous())
    if x0 is None:
        x = ops.zeros_like(bb)
        r = ops.lincomb([(1.0, bb)])                      # b - A(0): A is linear, skip the wasted operator application
    else:
        x = x0.reshape(bb.shape).contiguous().float().clone()
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
    b = b.contiguous()
    if b.dtype != torch.float32:
        b = b.float()
    bb, flat = _as_batch(b)
    B = bb.shape[0]
    apply = (lambda t: A(t.reshape(b.shape)).reshape(bb.shape).contiguous())
    if x0 is None:
        x = torch.zeros_like(bb)
        r = bb.clone()                      # b - A(0): A is linear, skip the wasted operator application
    else:
        x = x0.reshape(bb.shape).contiguous().float()
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
    pin = torch.empty((B, B), dtype=torch.float32, pin_memory=True) if on_gpu else None

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
