"""Fused ADMM iteration for recognised objective graphs.

``plan_admm`` pattern-matches the compiled problem:

    Omega : sum_squares( conv(x, psf) - b )  |  sum_squares( x - b )          (any number)
    Psi   : norm1 / norm2 / nonneg / deep_prior(FFDNet)  of  x | grad(x, 0) | grad(x, 1)   (<= 4 terms)

and replaces the reference's per-iteration graph walks (dprox/algo/admm.py:49-59 ->
proxfn/sum_square.py:123-156 -> linop/comp_graph.py:198-282; 18 full complex FFTs and ~80 eager ops
per TV-deconvolution iteration) by three stages of hand-written kernels working in place on
HBM-resident state:

    0. FK    = F(sum_Omega K^T b)  once per solve, fp64 transform       (dpx_data_spectrum)
    1. rhs   = rho * sum_i K_i^T (v_i - u_i)                           (dpx_admm_rhs: spatial stencils)
    2. x     = irFFT2[(rFFT2(rhs) + FK + eps) / (|H|^2 + rho sum|G_i|^2 + eps)]   (dpx_fourier_solve)
    3. v_i   = prox_i(K_i x + u_i) ; u_i += K_i x - v_i               (dpx_admm_zupdate [+ FFDNet])

K^T b and the denominators are computed once per solve; rho / lambda schedules are uploaded once as
[T, B] device arrays.  Anything that does not match returns ``None`` and the caller falls back to the
op-by-op path on the same HIP primitives (never to PyTorch or the CPU).
"""
import ctypes
import os

import torch
from tqdm import tqdm

from .. import _backend as be
from .. import _ops as ops
from . import autodiff
from ..linop import Constant, Variable, conv, conv_doe, grad
from ..linop import sum as lin_sum
from ..proxfn import deep_prior, least_squares, nonneg, norm1, norm2, sum_squares
from ..proxfn.pnp.denoisers import Denoiser2D, FFDNetColorDenoiser, FFDNetDenoiser


def _is_var(op):
    return isinstance(op, Variable)


def _is_const(op):
    """a Constant / Placeholder, or a subtree over constants only (``- placeholder`` is ``scale(-1, Placeholder)``)"""
    return isinstance(op, Constant) or len(op.variables) == 0


def _omega_ok(fn):
    """sum_squares over  x | conv(x) , optionally minus constants"""
    if type(fn) is not sum_squares or fn.beta != 1:
        return False
    op = fn.linop
    if isinstance(op, lin_sum):
        kids = list(op.input_nodes)
        lin = [k for k in kids if not _is_const(k)]
        if len(lin) != 1:
            return False
        op = lin[0]
    if _is_var(op):
        return True
    # (a linearised conv_doe pads / crops around its FFT product: only the op-by-op path reproduces that)
    return type(op) in (conv, conv_doe) and getattr(op, "circular", True) and _is_var(op.input_nodes[0])


def _omega_conv(fn):
    """the conv node of a recognised Omega term (None for the identity)"""
    op = fn.linop
    if isinstance(op, lin_sum):
        op = [k for k in op.input_nodes if not _is_const(k)][0]
    return op if type(op) in (conv, conv_doe) else None


def _psi_linop_code(op):
    if _is_var(op):
        return be.LIN_IDENTITY
    if type(op) is grad and _is_var(op.input_nodes[0]) and op.dim in (0, 1):
        return be.LIN_GRAD_H if op.dim == 0 else be.LIN_GRAD_W
    return None


def _psi_prox_code(fn):
    if fn.beta != 1:
        return None
    if type(fn) is norm1:
        return be.PROX_NORM1
    if type(fn) is nonneg:
        return be.PROX_NONNEG
    if type(fn) is norm2:
        return be.PROX_SUMSQ
    if type(fn) is deep_prior and not fn.unroll and not fn.clamp and isinstance(fn.denoiser, (FFDNetColorDenoiser, FFDNetDenoiser)):
        return be.PROX_EXTERNAL
    return None


def plan_fingerprint(solver):
    """everything mutable the pattern match of ``plan_admm`` reads, cheaply: the term lists (identity and type of every term), the
    terms' beta / unroll / clamp / x8 / sqrt, the denoiser's class AND identity, the linop at the root of each term (with the version of
    its tables where it has any: a kernel swapped in place) and the x-update's kind.  Part of the plan
    cache's key (``ADMM._plan_for``): a term swapped, re-weighted or re-configured after the first solve gets a new match instead
    of a stale plan."""
    ls = getattr(solver, "least_square", None)
    terms = []
    for fn in list(solver.psi_fns) + list(solver.omega_fns):
        op = fn.linop
        beta = fn.beta
        if isinstance(beta, torch.Tensor):                  # (a tensor-valued weight: by identity and version -- no device read-back on the
            beta = ("tensor", id(beta), beta._version)      #  plan-cache lookup of every solve; never `==` on tensors)
        den = getattr(fn, "denoiser", None)
        terms.append((id(fn), type(fn), beta, getattr(fn, "unroll", None), getattr(fn, "clamp", None), type(den), id(den),
                      getattr(fn, "x8", None), getattr(fn, "sqrt", None), id(op), type(op), getattr(op, "dim", None),
                      getattr(op, "tables_version", lambda: None)()))
    return (len(solver.psi_fns), tuple(terms), id(ls), bool(getattr(ls, "freq_diagonalizable", False)))


def plan_admm(solver, state):
    ls = getattr(solver, "least_square", None)
    if not isinstance(ls, least_squares) or not ls.freq_diagonalizable:
        return None
    psi, omega = list(solver.psi_fns), list(solver.omega_fns)
    if len(psi) > be.MAX_TERMS or not all(_omega_ok(fn) for fn in omega):
        return None
    codes = []
    for fn in psi:
        lc, pc = _psi_linop_code(fn.linop), _psi_prox_code(fn)
        if lc is None or pc is None:
            return None
        if pc == be.PROX_EXTERNAL and lc != be.LIN_IDENTITY:
            return None
        codes.append((lc, pc))
    x = state[0]
    if x.ndim != 4 or x.dtype != torch.float32:
        return None
    if len(solver.Kall.variables) != 1:
        return None
    return FusedADMM(solver, codes)


def lazy_initial_state(solver, x0, plan):
    """ADMM.solve's private initial state for problems the two-kernel iteration takes: (x0, [v_i], [u_i]) whose split variables are
    ALLOCATED but not computed -- v_i = K_i x0 and u_i = 0 are implied (``solver._fresh`` says so) and the fresh-state path never
    reads them: the seed pass forms the first right-hand side from x0, the first iteration counts the duals as zero (only row 0 of
    plane 0 of every image of every u_i is zeroed: the rows it fetches), every v_i / u_i is fully written before the solve returns.  Saves the two
    K_i x0 passes and the two zero fills of ``initialize`` (0.17 ms at 8 x 3 x 1024^2).  ``materialize_state`` turns it into the
    real thing for any path that does read the state.  None when the problem does not qualify."""
    if not (isinstance(x0, torch.Tensor) and x0.ndim == 4 and x0.dtype == torch.float32 and x0.is_contiguous()):
        return None
    if plan is None or not plan.codes or any(pc == be.PROX_EXTERNAL for _, pc in plan.codes):
        return None
    B, C, H, W = x0.shape
    early = ops.make_terms([dict(linop=lc, prox=pc, alpha=1.0, lam=None, v=x0, u=x0) for lc, pc in plan.codes])
    if not ops.iter_supported(H, W, early, len(plan.codes)):
        return None
    n = len(plan.codes)
    v = [torch.empty_like(x0) for _ in range(n)]
    U = torch.empty((n,) + tuple(x0.shape), dtype=x0.dtype, device=x0.device)
    U[:, :, 0, 0, :].zero_()                                   # (row 0 of plane 0 of every IMAGE: sub-batch chains start at any of them; one
    u = list(U.unbind(0))                                      #  fill for all terms -- a training step of 1 ms counts its 4-us launches)
    solver._fresh = (x0, v, u, [t._version for t in [x0] + v + u])
    solver._fresh_lazy = True
    return x0, v, u


def materialize_state(solver, state):
    """a lazy initial state (above) becomes what ``ADMM.initialize`` returns: v_i = K_i x0, u_i = 0, written into the same tensors"""
    if not getattr(solver, "_fresh_lazy", False):
        return
    solver._fresh_lazy = False
    if not fresh_state(solver, state):
        return
    x0, v, u = state
    kx = solver.K.forward(x0, return_list=True) or []
    for dst, src in zip(v, kx):
        dst.copy_(src)
    for t in u:
        t.zero_()
    solver._fresh = (x0, v, u, [t._version for t in [x0] + v + u])


def fresh_state(solver, state):
    """True when ``state`` is exactly what ``ADMM.initialize`` returned last (same tensors, never written since): v_i = K_i x0 and
    u_i = 0, so the first right-hand side can be formed from x0 alone and the first iteration need not stream the (zero) duals"""
    f = getattr(solver, "_fresh", None)
    if f is None or len(state) != 3:
        return False
    x, v, u, vers = f
    sx, sv, su = state
    if sx is not x or len(sv) != len(v) or len(su) != len(u):
        return False
    if any(a is not b for a, b in zip(list(sv) + list(su), v + u)):
        return False
    return all(t._version == n and t.is_contiguous() for t, n in zip([x] + v + u, vers))


_sched_cache = {}        # (data_ptr, shape, version, T, B, device) -> (source tensor, table): a schedule handed in again costs nothing


def schedule_table(vals, T, B, device):
    """0-d / [T] / [B,T] -> contiguous [T,B] float32 on device (cached while the caller passes the same, unmodified tensor: a loop
    that calls iters() / solve() with its own schedule tensors then launches no kernel for them)"""
    key = None
    # (device tensors only: a CPU tensor may share its memory with a NumPy array, whose edits do not bump the version counter)
    if isinstance(vals, torch.Tensor) and not vals.requires_grad and vals.is_cuda:
        key = (vals.data_ptr(), tuple(vals.shape), tuple(vals.stride()), vals._version, vals.dtype, str(vals.device), T, B, str(device))
        hit = _sched_cache.get(key)
        if hit is not None:
            return hit[1]
    v = vals.to(device=device, dtype=torch.float32)
    if v.ndim == 0:
        v = v.expand(T)
    if v.ndim == 1:
        tab = v[:T].reshape(T, 1).expand(T, B).contiguous()
    elif v.ndim == 2 and v.shape[0] == B:
        tab = v[:, :T].t().contiguous()
    else:
        raise be.DpxError(f"schedule of shape {tuple(vals.shape)} does not fit batch {B} x {T} iterations")
    if key is not None:
        if len(_sched_cache) >= 32:
            _sched_cache.pop(next(iter(_sched_cache)))
        _sched_cache[key] = (vals, tab)          # (the source is kept alive: its address cannot be handed to another tensor meanwhile)
    return tab


def _sigma_table(fn, lt):
    """[T, B] noise levels of a deep_prior term from its lambda schedule: alpha * lam, or safe_sqrt(alpha * lam) with sqrt=True
    (the scaling of `c * deep_prior(...)` enters before the root: prior.py:77 behind ProxFn.prox, proxfn/base.py:55-64)"""
    lt = lt * float(fn.alpha) if float(fn.alpha) != 1.0 else lt
    return torch.sqrt(torch.clamp(lt, min=1e-8)) if fn.sqrt else lt


def _denoise_split(fn, d, sig):
    """z-update of a deep_prior term: v = D(d; sigma) with d = x + u (the denoiser's own HIP kernels)"""
    B, C, H, W = d.shape
    den = fn.denoiser
    if isinstance(den, Denoiser2D):
        return den.model(d.reshape(B * C, 1, H, W), sig.repeat_interleave(C) if C > 1 else sig).reshape(B, C, H, W)
    return den.model(d, sig)


def plan_split_cg(solver, state, rhos, lams):
    """ADMM / LinearizedADMM whose x-update is a CG solve (the stacked operator is not diagonalisable: config 4's subsampled
    Fourier data term) and whose Psi terms all act on x itself.  Returns a FusedSplitCG or None."""
    ls = getattr(solver, "least_square", None)
    if not isinstance(ls, least_squares) or ls.diagonalizable or ls.freq_diagonalizable:
        return None
    if ls.linear_solve_config.solver_type != "cg" or ls.linear_solve_config.verbose:
        return None
    psi = list(solver.psi_fns)
    if not psi or len(psi) > be.MAX_TERMS:
        return None
    codes = []
    for fn in psi:
        lc, pc = _psi_linop_code(fn.linop), _psi_prox_code(fn)
        if lc != be.LIN_IDENTITY or pc is None:
            return None
        codes.append((lc, pc))
    x = state[0]
    if x.ndim != 4 or x.dtype != torch.float32 or len(solver.Kall.variables) != 1:
        return None
    tensors = [x, rhos] + list(lams.values()) + list(state[1]) + list(state[2])
    if torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors):
        return None
    return FusedSplitCG(solver, codes)


class FusedSplitCG:
    """One iteration = three stages, every one a HIP kernel sequence issued without host synchronisation:
         rhs  : Ktb = sum_Omega K^T b + rho sum_i (v_i - u_i)                       (dpx_admm_rhs, one pass)
         x    : CG on (sum_Omega K^T K + n rho I) x = Ktb, controlled on the device   (linalg.solve.cg -> dpx_cg_*)
         z    : d_i = x + u_i, v_i = prox_i(d_i), u_i = d_i - v_i                    (dpx_admm_zupdate [+ the denoiser])
    ADMM (admm.py:49-59) and LinearizedADMM (admm.py:78-100) coincide here: with K_i = I the linearised right-hand side
    x - (x - v_i + u_i) is v_i - u_i (the reference evaluates the former in fp32; the difference is one rounding of x, ~6e-8
    relative, far below the 1e-5 parity bar and pinned by fixtures G7 / G32)."""

    def __init__(self, solver, codes):
        self.solver, self.codes = solver, codes

    def run(self, state, rhos, lams, max_iter, pbar=False, callback=None):
        s = self.solver
        ls = s.least_square
        psi = list(s.psi_fns)
        x0, v, u = state
        B, C, H, W = x0.shape
        dev = x0.device
        T = max_iter
        s.Kall.update_vars([x0])
        if T <= 0:
            return state
        rho_tab = schedule_table(rhos, T, B, dev)
        lam_tab = []
        for fn in psi:
            lt = schedule_table(lams[fn], T, B, dev)
            lam_tab.append(_sigma_table(fn, lt) if isinstance(fn, deep_prior) else lt)
        v = [t.contiguous() for t in v]
        u = [t.contiguous() for t in u]
        x = x0
        rhs = torch.empty_like(x0)
        specs = [dict(linop=lc, prox=pc, alpha=float(fn.alpha), lam=lam_tab[i][0], v=v[i], u=u[i])
                 for i, (fn, (lc, pc)) in enumerate(zip(psi, self.codes))]
        terms = ops.make_terms(specs)
        n = len(specs)
        ext = [i for i, (_, pc) in enumerate(self.codes) if pc == be.PROX_EXTERNAL]
        var = s.Kall.variables[0]
        ktb = ls.quad_rhs()
        if ktb is not None and ktb.shape != x0.shape:
            ktb = ktb.expand_as(x0).contiguous()
        # one gray FFDNet prior on single-channel images behind the masked-Fourier CG (config 4): the whole iteration is ONE C call
        # (dpx_admm_cg_pnp_iter) -- the host language's layers between the CG solve's stop flag and the denoiser's first launch were 44 us of
        # idle stream per iteration of a 4 x 1 x 320^2 shard.  DPX_SPLIT_CG_STAGED=1: the stage-by-stage loop below (A/B, tests).
        cfg = ls.linear_solve_config
        sysm = ls._masked_fft_system(False) if (cfg.solver_type == "cg" and not cfg.verbose) else None
        one_call = (len(ext) == 1 and C == 1 and isinstance(psi[ext[0]].denoiser, FFDNetDenoiser) and psi[ext[0]].denoiser.model.in_nc == 1
                    and not torch.is_grad_enabled() and sysm is not None and B <= 64 and not be.host_mode_skip_fast_cg()
                    and ls._masked_fft_fits(sysm[0], x0) and not os.environ.get("DPX_SPLIT_CG_STAGED"))
        s.last_split_cg_loop = "one call" if one_call else "staged"     # (tests / tools: which loop the last solve took)
        if one_call:
            e = ext[0]
            x = torch.empty_like(x0)
            v_new = torch.empty_like(x0)
            net = psi[e].denoiser.model
            net_key = lambda: (net._weights_version(), net.compute_mode)
            step, step_key = ops.CgPnpIter(x, rhs, ktb, terms, n, e, sysm[0], sysm[1], cfg.rtol, cfg.max_iters, net), net_key()
            hist, run_hist = list(getattr(s, "_cg_exit_hist", ())), []
            s._cg_exit_hist = run_hist
            xs = (x, torch.empty_like(x0)) if step.folds else (x, x)  # (folded tail: iteration t + 1's iterate is zeroed while x_t is still the result)
            for it in tqdm(range(T), disable=not pbar):
                for i in range(n):
                    terms[i].lam = lam_tab[i][it].data_ptr()
                x = xs[it & 1]
                hint = hist[it] if it < len(hist) else -1            # (where the same solve of this solver's previous run ended)
                if step.folds and it + 1 < T and callback is None:     # (a callback may solve something else in between: the prepared CG state lives in a shared workspace)
                    n_cg = step(x, v_new, rho_tab[it], lam_tab[e][it], rho_tab[it + 1], xs[(it + 1) & 1], cg_hint=hint)
                else:
                    n_cg = step(x, v_new, rho_tab[it], lam_tab[e][it], cg_hint=hint)
                ls.cg_iters.append(n_cg)
                run_hist.append(n_cg)
                v[e], v_new = v_new, v[e]                            # the denoised image becomes v; its old buffer is the next target
                terms[e].v = v[e].data_ptr()
                var.value = x
                if callback is not None:
                    s._notify_all_op_current_step(it)
                    callback(iter=it, state=(x, v, u), rho=rhos[..., it], lam={k: val[..., it] for k, val in lams.items()})
                    if net_key() != step_key:                        # the callback touched the denoiser (weights / arithmetic mode): resolve the
                        step, step_key = ops.CgPnpIter(xs[0], rhs, ktb, terms, n, e, sysm[0], sysm[1], cfg.rtol, cfg.max_iters, net), net_key()   # packed weights again
            s.Kall.update_vars([x])
            return x, v, u
        for it in tqdm(range(T), disable=not pbar):
            for i in range(n):
                terms[i].lam = lam_tab[i][it].data_ptr()
            ops.admm_rhs(rhs, ktb, rho_tab[it], terms, n)
            x = ls.solve_cg_rhs(rhs, rho_tab[it])
            ops.admm_zupdate(x, terms, n)
            for i in ext:                                            # v_i holds d = x + u_i
                d = v[i]
                out = _denoise_split(psi[i], d, lam_tab[i][it])
                ops.lincomb([(1.0, d), (-1.0, out)], out=u[i])       # u_i = d - v_i
                v[i] = out
                terms[i].v = out.data_ptr()
            var.value = x
            if callback is not None:
                s._notify_all_op_current_step(it)
                callback(iter=it, state=(x, v, u), rho=rhos[..., it], lam={k: val[..., it] for k, val in lams.items()})
        s.Kall.update_vars([x])
        return x, v, u


_TRACE = bool(os.environ.get("DPX_TRACE_HOST"))      # tuning: host-side time stamps of run() (microseconds since entry) on stderr
_trace_t0 = [0.0]


def _tr(tag):
    if _TRACE:
        import sys
        import time
        now = time.perf_counter()
        if tag == "enter":
            _trace_t0[0] = now
        sys.stderr.write(f"[dpx host] {tag:18s} {1e6 * (now - _trace_t0[0]):8.1f} us\n")


_CHAIN_STREAMS = {}          # device index -> candidate side streams of the sub-batch chains
_CHAIN_CHOICE = {}           # (device index, caller's stream, n) -> the n side streams that overlap with it (None: there are none)
_chain_spec_bytes = {}       # (B, C, H, W, chains) -> bytes of one spectrum buffer per chain
_chain_tab_cache = {}        # (id(table), b0, b1) -> (table, version, contiguous [T, b1 - b0] copy): the schedule tables are cached objects themselves


def _chain_table(tab, b0, b1):
    """columns [b0, b1) of a [T, B] schedule table as a contiguous table of their own (cached while the source table lives unchanged)"""
    key = (id(tab), b0, b1)
    hit = _chain_tab_cache.get(key)
    if hit is not None and hit[0] is tab and hit[1] == tab._version:
        return hit[2]
    out = tab[:, b0:b1].contiguous()
    if len(_chain_tab_cache) > 64:
        _chain_tab_cache.clear()
    _chain_tab_cache[key] = (tab, tab._version, out)
    return out


def _concurrent_side_streams(dev, main_handle, n):
    """n streams of the device whose kernels overlap with the caller's stream and with each other, or None.  HIP multiplexes streams onto
    a few hardware queues (GPU_MAX_HW_QUEUES, default 4) in creation order and streams on one queue run strictly one after the other:
    whether a new stream shares the caller's queue depends on what else created streams before (with RCCL initialised the first side
    stream did: the chains serialised, 5500 -> 4300 it/s).  So the streams are CHOSEN: candidates are probed with dpx_streams_concurrent
    (~1 ms each, drains the streams) once per (device, caller's stream) and the choice is kept."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, main_handle, n)
    if key in _CHAIN_CHOICE:
        return _CHAIN_CHOICE[key]
    pool = _CHAIN_STREAMS.setdefault(idx, [])
    L = be.lib()
    chosen = []
    for k in range(8):                                        # (at most 8 candidates: twice the default number of hardware queues)
        if k >= len(pool):
            pool.append(torch.cuda.Stream(device=dev))
        cand = pool[k]
        others = [main_handle] + [c.cuda_stream for c in chosen]
        if all(L.query("dpx_streams_concurrent", ctypes.c_void_p(o), ctypes.c_void_p(cand.cuda_stream)) == 1 for o in others):
            chosen.append(cand)
            if len(chosen) == n:
                break
    _CHAIN_CHOICE[key] = chosen if len(chosen) == n else None
    return _CHAIN_CHOICE[key]


def chain_streams(dev, chains):
    """(the caller's current stream, chains - 1 side streams that overlap with it) of a device, or None when the device has no such
    streams left (the caller then runs one chain)"""
    main = torch.cuda.current_stream(dev)
    side = _concurrent_side_streams(dev, main.cuda_stream, chains - 1)
    return None if side is None else (main, list(side))


def chain_stream_handles(dev, chains):
    """raw handles [caller's current stream, side streams ...] for the C ABI; [None] * chains on the CPU emulator (the chains then
    run one after the other); None when no overlapping side streams exist"""
    if be.host_mode():
        return [None] * chains
    h = be.stream().value or 0                                # (0: the null stream)
    side = _concurrent_side_streams(dev, h, chains - 1)
    return None if side is None else [h] + [st.cuda_stream for st in side]


def chain_bounds(B, chains, c):
    """images [b0, b1) of sub-batch chain c"""
    return (c * B) // chains, ((c + 1) * B) // chains


def sub_batch_chains(B, C, H, W):
    """how many independent sub-batch chains the two-kernel iteration of a [B,C,H,W] problem is run as (FusedADMM._run_chains): 2 when
    each half still fills the GPU by itself (measured on 2 ... 16 images of 1 ... 3 x 512^2 ... 1024^2, tools/concurrent_probe.py,
    tools/odd_probe.py; small problems are latency-bound and stay one chain).
    DPX_CHAINS=n forces n (1 = off)."""
    env = os.environ.get("DPX_CHAINS")
    if env:
        n = max(1, int(env))
        return n if B >= n else 1
    if B >= 2 and W in (256, 512, 1024) and (B // 2) * C * H * W >= (3 << 20):      # (odd batches split unevenly: 3 / 5 / 7 images of 3x1024^2 gain 10 - 13 % too)
        return 2
    return 1


class FusedADMM:
    merge_z_rhs = True      # staged iteration (planes off the two-kernel iteration): z / dual stage + next right-hand side as one pass

    def __init__(self, solver, codes):
        self.solver, self.codes = solver, codes

    def run(self, state, rhos, lams, max_iter, pbar=False, callback=None, dual=True, vxu=False):
        """_run, and on an exception behind the chains' seed launches the caller's stream joins the side streams before the chains' buffers
        are released (a block handed back to the allocator while a side stream still writes it would be given out again)"""
        self._pending_chains = None
        try:
            return self._run(state, rhos, lams, max_iter, pbar, callback, dual, vxu)
        except BaseException:
            pre = self._pending_chains
            if pre is not None:
                try:
                    ops.stream_join(pre["handles"][0], pre["handles"])
                except Exception:
                    pass
            raise
        finally:
            self._pending_chains = None

    def _run(self, state, rhos, lams, max_iter, pbar=False, callback=None, dual=True, vxu=False):
        """``dual=False``: half-quadratic splitting (hqs.py:4-20) = the same three stages with the dual variables pinned to
        zero -- state (x, [z_i]); the z-stage's ``u_out`` goes to a scratch buffer and is never read.
        ``vxu=True``: ADMM in the order v, x, u (admm.py:103-120) on the same stages: with u' = -u the split update is
        prox(K z + u') (z stage, its dual output discarded), the x-update sees v - u', and u' <- u' - v + z is one AXPY."""
        _tr("enter")
        s = self.solver
        ls = s.least_square
        psi = list(s.psi_fns)
        if dual:
            x0, v, u = state
        else:
            x0, v = state
            zero = torch.zeros_like(x0)
            u = [zero for _ in v]
        B, C, H, W = x0.shape
        dev = x0.device
        T = max_iter
        s.Kall.update_vars([x0])
        if T <= 0:                                           # nothing to do: the state is returned as it came
            materialize_state(s, state)
            return state

        raw_offs = [self._offset_autograd(fn, x0) for fn in s.omega_fns]
        trained_psfs = [cv.psf for cv in map(_omega_conv, s.omega_fns)
                        if isinstance(cv, conv_doe) and cv.circular and isinstance(cv.psf, torch.Tensor) and cv.psf.requires_grad]
        want_grad = dual and not vxu and autodiff.needs_grad(x0, rhos, lams, raw_offs, list(v) + list(u) + trained_psfs)
        # (the differentiable path builds its own schedule tables as autograd nodes, autodiff.run: none are made here for it)
        rho_tab = None if want_grad else schedule_table(rhos, T, B, dev)
        _tr("rho table")
        # The two-kernel iteration starts from the row-transformed right-hand side rho_0 sum K_i^T (v_i - u_i): that pass needs
        # nothing but the state, so it is launched FIRST and the rest of the host-side preparation (schedule tables, data spectrum,
        # denominators, workspaces: ~0.1 ms) runs while the GPU is already busy instead of in front of it.
        _tr("grad checks")
        seeded = None
        chains = 1
        if callback is None and not pbar and not vxu and torch.is_tensor(x0) and (x0.is_cuda or be.host_mode()):
            chains = sub_batch_chains(B, C, H, W)
            if chains > 1 and chain_stream_handles(dev, chains) is None:      # (no second hardware queue to run on)
                chains = 1
        fresh = dual and not vxu and fresh_state(s, state)
        lazy = fresh and getattr(s, "_fresh_lazy", False)
        # the differentiable path may keep a lazy state when nothing is differentiated THROUGH the state and the whole loop is one C call on
        # the two-kernel iteration (autodiff._UnrolledClosed): the C side then forms the first right-hand side from x0 (dpx_admm_rhs_fresh)
        grad_fresh = (lazy and want_grad and T > 0 and len(psi) > 0 and not x0.requires_grad and not trained_psfs
                      and all(pc != be.PROX_EXTERNAL for _, pc in self.codes) and not os.environ.get("DPX_UNROLL_CHAIN"))
        if lazy and not grad_fresh and (want_grad or T <= 0 or len(psi) == 0):     # (a path that reads the split variables)
            materialize_state(s, state)
            lazy = False
        s._fresh = None                                        # (the state is about to be advanced in place)
        s._fresh_lazy = False
        if not want_grad and not vxu and len(psi) > 0 and all(pc != be.PROX_EXTERNAL for _, pc in self.codes):
            v = [t.contiguous() for t in v]
            u = [t.contiguous() for t in u]
            early = ops.make_terms([dict(linop=lc, prox=pc, alpha=float(fn.alpha), lam=None, v=v[i], u=u[i])
                                    for i, (fn, (lc, pc)) in enumerate(zip(psi, self.codes))])
            if ops.iter_supported(H, W, early, len(psi)):
                if chains > 1:                                     # (every chain seeds its own spectrum buffer on its own stream, now)
                    seeded = self._seed_chains(x0, v, u, rho_tab, fresh, chains, dev)
                else:
                    seeded = ops.admm_seed_rows(ops.spectrum_buffer(B * C, H, W, dev), rho_tab[0], early, len(psi), x0.shape, dev,
                                                fresh_x=x0 if fresh else None)
        _tr("seeds issued")
        if not isinstance(seeded, dict):
            chains = 1
        if lazy and seeded is None and not grad_fresh:         # (cannot happen for a state lazy_initial_state handed out; kept as a guard)
            s._fresh, s._fresh_lazy = (x0, list(v), list(u), [t._version for t in [x0] + list(v) + list(u)]), True
            materialize_state(s, state)
            s._fresh, fresh = None, False
        lam_tab = []
        for fn in (() if want_grad else psi):
            lt = schedule_table(lams[fn], T, B, dev)
            lam_tab.append(_sigma_table(fn, lt) if isinstance(fn, deep_prior) else lt)     # deep priors: the table holds sigma
        # data spectrum F(sum_Omega K^T b): fp64 transform, kept in the Fourier domain, recomputed only when an
        # offset (the observation b) changes
        _tr("lam tables")
        FK = self._data_spectrum(x0, chains)
        _tr("data spectrum")
        (t0, c0), (t1, c1) = ls.diag_tables(x0.shape, dev, True)
        _tr("diag tables")

        # ---- differentiable (unrolled-training) mode: hand-written backward stages, autodiff.py -------------------
        if want_grad:
            otfs, doe = [], []
            for fn in s.omega_fns:
                cv = _omega_conv(fn)
                otfs.append(cv._tables(x0.shape, dev) if cv is not None else None)
                if isinstance(cv, conv_doe) and cv.circular and isinstance(cv.psf, torch.Tensor) and cv.psf.requires_grad and torch.is_grad_enabled():
                    # end-to-end optics: the PSF is trained through the solver -- its OTF enters the x-updates as an autograd tensor
                    from ..linop.fourier import _doe_padded
                    P = _doe_padded(cv.psf.float().to(dev), x0.shape).expand(1, C, H, W).contiguous()
                    off = fn.offset
                    Yhat = None if off is None else ops.cfft2(off.detach().to(dev).expand_as(x0).contiguous(), inverse=False, centred=False, ortho=False)
                    doe.append((autodiff._FullOtf.apply(P), Yhat))
            plan = autodiff.DiffPlan(self.codes, psi, (t0, c0, t1, c1), FK, otfs, ls_eps(ls), hist_bf16=getattr(s, "unroll_dtype", "f32") == "bf16",
                                     doe=doe)
            diff_offs = [o if o is not None else ops.zero_scalar(dev) for o in raw_offs]
            if grad_fresh and not doe:
                plan.fresh_x0 = x0
            elif grad_fresh:                                       # (a trained PSF: the stage-by-stage path reads the state)
                s._fresh, s._fresh_lazy = (x0, list(v), list(u), [t._version for t in [x0] + list(v) + list(u)]), True
                materialize_state(s, state)
                s._fresh = None
            x, v, u = autodiff.run(plan, (x0, v, u), rhos, {fn: lams[fn] for fn in psi}, T, diff_offs)
            s.Kall.update_vars([x.detach()])
            return x, v, u

        v = [t.contiguous() for t in v]
        u = [t.contiguous() for t in u]
        x = torch.empty_like(x0)
        rhs = None if seeded is not None else torch.empty_like(x0)        # (the two-kernel iteration never forms the right-hand side as an image)
        specs = [dict(linop=lc, prox=pc, alpha=float(fn.alpha), lam=lam_tab[i][0], v=v[i], u=u[i])
                 for i, (fn, (lc, pc)) in enumerate(zip(psi, self.codes))]
        terms = ops.make_terms(specs)
        n = len(specs)
        ext = [i for i, (_, pc) in enumerate(self.codes) if pc == be.PROX_EXTERNAL]
        var = s.Kall.variables[0]

        if not dual or vxu:
            scratch = torch.empty_like(x0)
            for i in range(n):
                terms[i].u_out = scratch.data_ptr()
        if vxu:
            x.copy_(x0)
            for i in range(n):                                   # u' = -u, kept in fresh buffers
                u[i] = ops.lincomb([(-1.0, u[i])])
                terms[i].u = u[i].data_ptr()
            if callback is None and not pbar and T >= 3 and not ext and n > 0 and ops.iter_supported(H, W, terms, n):
                return self._run_vxu_two_kernel(x0.shape, dev, T, terms, n, v, u, x, FK, (t0, c0, t1, c1), rho_tab, lam_tab)
            for it in tqdm(range(T), disable=not pbar):
                for i in range(n):
                    terms[i].lam = lam_tab[i][it].data_ptr()
                if n > 0 and not ext and self.merge_z_rhs:           # v-update and right-hand side of the same iteration: one pass (the duals
                    ops.admm_zupdate_rhs(x, terms, n, rhs, rho_tab[it], dual=False)      # it would write go to scratch: rhs from the incoming ones)
                else:
                    ops.admm_zupdate(x, terms, n)
                    ops.admm_rhs(rhs, None, rho_tab[it], terms, n)
                ops.fourier_solve(rhs, t0, t1, c0, c1, rho_tab[it], ls_eps(ls), out=x, spec_add=FK)
                for i in range(n):
                    ops.lincomb([(1.0, u[i]), (-1.0, v[i]), (1.0, x)], out=u[i])
                var.value = x
                if callback is not None:
                    s._notify_all_op_current_step(it)
                    callback(iter=it, state=(x, v, [ops.lincomb([(-1.0, t)]) for t in u]), rho=rhos[..., it],
                             lam={k: val[..., it] for k, val in lams.items()})
            s.Kall.update_vars([x])
            return x, v, [ops.lincomb([(-1.0, t)]) for t in u]
        if not ext and n > 0 and ops.iter_supported(H, W, terms, n):
            if not dual:                                         # half-quadratic splitting: the same two kernels with the duals counted as zero
                for i in range(n):
                    terms[i].reserved = be.TERM_NO_DUAL
            _tr("terms built")
            if chains > 1:
                return self._run_chains(x0, dev, T, n, v, u, x, FK, (t0, c0, t1, c1), rho_tab, lam_tab, dual, fresh, chains, seeded)
            return self._run_two_kernel(x0.shape, dev, T, terms, n, v, u, x, rhs, FK, (t0, c0, t1, c1), rho_tab, lam_tab,
                                        rhos, lams, pbar, callback, dual, seeded, fresh and seeded is not None)

        # one FFDNet prior, everything else closed-form: the whole iteration is ONE C call (dpx_admm_pnp_iter)
        one_call = (dual and len(ext) == 1 and isinstance(psi[ext[0]].denoiser, (FFDNetColorDenoiser, FFDNetDenoiser))
                    and not torch.is_grad_enabled() and psi[ext[0]].denoiser.model.in_nc in (C, 1))
        if one_call:
            e = ext[0]
            net = psi[e].denoiser.model
            gray = net.in_nc != C
            sig_tab = lam_tab[e].repeat_interleave(C, dim=1).contiguous() if (gray and C > 1) else lam_tab[e]
            dd = ops.denominator(t0, c0, t1, c1, C, H, W, dev)
            v_new = torch.empty_like(x0)
            for it in tqdm(range(T), disable=not pbar):
                for i in range(n):
                    terms[i].lam = lam_tab[i][it].data_ptr()
                ops.admm_pnp_iter(x, rhs, terms, n, e, v_new, rho_tab[it], sig_tab[it], FK, dd, ls_eps(ls), net)
                v[e], v_new = v_new, v[e]                            # the denoised image becomes v; its old buffer is the next target
                terms[e].v = v[e].data_ptr()
                var.value = x
                if callback is not None:
                    s._notify_all_op_current_step(it)
                    callback(iter=it, state=(x, v, u), rho=rhos[..., it], lam={k: val[..., it] for k, val in lams.items()})
            s.Kall.update_vars([x])
            return x, v, u

        # closed-form proxes only: the z / dual stage of iteration t and the right-hand side of iteration t + 1 are ONE pass
        # (dpx_admm_zupdate_rhs: 8 instead of 12 plane passes; it reads duals at neighbouring pixels, so the duals alternate between two
        # sets of buffers -- half-quadratic splitting's are write-only scratch already)
        merged = n > 0 and not ext and self.merge_z_rhs
        if merged and not dual:                                      # half-quadratic splitting: the duals are zero -- the merged pass need not fetch them
            for i in range(n):
                terms[i].reserved = be.TERM_NO_DUAL
        u_given = list(u)                                            # the caller's dual tensors
        if merged and dual:
            u_alt = [torch.empty_like(t) for t in u]
            for i in range(n):
                terms[i].u_out = u_alt[i].data_ptr()
        for it in tqdm(range(T), disable=not pbar):
            for i in range(n):
                terms[i].lam = lam_tab[i][it].data_ptr()
            if not merged or it == 0:
                ops.admm_rhs(rhs, None, rho_tab[it], terms, n)
            ops.fourier_solve(rhs, t0, t1, c0, c1, rho_tab[it], ls_eps(ls), out=x, spec_add=FK)
            if merged and it + 1 < T:
                ops.admm_zupdate_rhs(x, terms, n, rhs, rho_tab[it + 1], dual=dual, emit_v=callback is not None)   # (v: only a callback looks at it before the last stage)
            else:
                ops.admm_zupdate(x, terms, n)
            if merged and dual:
                for i in range(n):                                   # the duals just written become the next iteration's input
                    u[i], u_alt[i] = u_alt[i], u[i]
                    terms[i].u, terms[i].u_out = u[i].data_ptr(), u_alt[i].data_ptr()
            for i in ext:                                            # v_i holds d = x + u_i
                d = v[i]
                out = _denoise_split(psi[i], d, lam_tab[i][it])
                ops.lincomb([(1.0, d), (-1.0, out)], out=u[i])       # u_i = d - v_i
                v[i] = out
                terms[i].v = out.data_ptr()
            var.value = x
            if callback is not None:
                s._notify_all_op_current_step(it)
                callback(iter=it, state=(x, v, u) if dual else (x, v), rho=rhos[..., it], lam={k: val[..., it] for k, val in lams.items()})
        if merged and dual:
            # the duals alternated between the caller's tensors and u_alt: after an odd number of iterations the current ones sit in
            # u_alt -- they go back into the tensors that came in, so that a caller holding on to its state's duals sees this call's
            # result there (as with the in-place dual update of the un-merged stages), and the returned state IS those tensors
            for i in range(n):
                if u[i] is not u_given[i]:
                    u_given[i].copy_(u[i])
                    u[i] = u_given[i]
        s.Kall.update_vars([x])
        return (x, v, u) if dual else (x, v)


    def _run_vxu_two_kernel(self, shape, dev, T, terms, n, v, up, x, FK, diag, rho_tab, lam_tab):
        """ADMM in the order v, x, u (admm.py:103-120) on the two-kernel iteration (power-of-two planes, closed-form proxes, no callback).
        Per iteration j the reference does  v_i = prox(K_i z_j - u_i);  z_{j+1} = solve(v_i + u_i);  u_i += v_i - z_{j+1}  (z itself,
        not K_i z: the reference's dual, kept).  With u' = -u and q_i = u'_i - v_i the row pass behind solve j forms the dual
        u'_i = q_i + z_{j+1}, the NEXT v-update v_i = prox(K_i z_{j+1} + u'_i), q'_i = u'_i - v_i and the next right-hand side
        rho_{j+1} sum K_i^T (v_i - u'_i): one plane in, one plane out per term like ADMM (DPX_TERM_VXU).  The first v-update and the
        last x-update / dual are the staged kernels (the loop is one v-update ahead of the reference's iteration boundaries).
        up: the negated duals u'.  Returns (z, [v_i], [u_i])."""
        s = self.solver
        ls = s.least_square
        B, C, H, W = shape
        t0, c0, t1, c1 = diag
        var = s.Kall.variables[0]
        scratch = torch.empty_like(x)
        # v-update of iteration 0 (dpx_admm_zupdate: v = prox(K x + u'); its dual output is not used)
        for i in range(n):
            terms[i].lam = lam_tab[i][0].data_ptr()
            terms[i].u_out = scratch.data_ptr()
        ops.admm_zupdate(x, terms, n)
        # seed: row transform of rho_0 sum K^T (v - u'), then q = u' - v in place of u'
        chains = sub_batch_chains(B, C, H, W) if (x.is_cuda or be.host_mode()) else 1
        if chains > 1 and chain_stream_handles(dev, chains) is None:
            chains = 1
        dd = ops.denominator(t0, c0, t1, c1, C, H, W, dev)
        if chains > 1:
            # sub-batch chains (_run_chains): every chain seeds from its images of (v, u') and walks its own spectrum buffers
            pre = self._seed_chains(x, v, up, rho_tab, False, chains, dev)
            q_cur = [ops.lincomb([(1.0, up[i]), (-1.0, v[i])]) for i in range(n)]
            q_nxt = [torch.empty_like(t) for t in q_cur]
            FKc = self._data_spectrum(x, chains)
            for wk in pre["work"]:
                b0, b1 = wk["b0"], wk["b1"]
                for i in range(n):
                    tm = wk["terms"][i]
                    tm.v, tm.u, tm.u_out = v[i][b0:b1].data_ptr(), q_cur[i][b0:b1].data_ptr(), q_nxt[i][b0:b1].data_ptr()
                    tm.reserved = be.TERM_VXU
                wk["lam"] = [_chain_table(lt, b0, b1)[1:] for lt in lam_tab]      # (shifted by one row, see below)
            L = be.lib()
            L.call("dpx_admm_iter_share", chains)
            try:
                ops.stream_fork(pre["handles"][0], pre["handles"])
                par = ops.admm_run_chains([dict(spec_a=wk["SA"], spec_b=wk["SB"], spec_add=fk, terms=wk["terms"], rho_tab=wk["rho"], lam_tabs=wk["lam"],
                                                x_out=x[wk["b0"]:wk["b1"]], B=wk["b1"] - wk["b0"], stream=wk["stream"]) for wk, fk in zip(pre["work"], FKc)],
                                          dd, n, ls_eps(ls), 0, T - 1, T, 1, shape, dev)
                ops.stream_join(pre["handles"][0], pre["handles"])
            finally:
                L.call("dpx_admm_iter_share", 1)
        else:
            SA = ops.spectrum_buffer(B * C, H, W, dev)
            SB = ops.spectrum_buffer(B * C, H, W, dev)
            ops.admm_seed_rows(SA, rho_tab[0], terms, n, shape, dev)
            q_cur = [ops.lincomb([(1.0, up[i]), (-1.0, v[i])]) for i in range(n)]
            q_nxt = [torch.empty_like(t) for t in q_cur]
            for i in range(n):
                terms[i].u, terms[i].u_out, terms[i].v = q_cur[i].data_ptr(), q_nxt[i].data_ptr(), v[i].data_ptr()
                terms[i].reserved = be.TERM_VXU
            # fused passes behind solves 0 .. T-2: the pass behind solve j uses lambda_{j+1} and rho_{j+1} (tables shifted by one row);
            # the last of them emits z_{T-1} and v of iteration T-1
            lam_shift = [lt[1:] for lt in lam_tab]
            par = ops.admm_run(SA, SB, FK, dd, terms, n, rho_tab, lam_shift, ls_eps(ls), 0, T - 1, T, x, True, shape, dev)
        q_fin = q_nxt if par else q_cur
        for i in range(n):
            terms[i].reserved = 0
        # iteration T-1's x-update and dual on the staged kernels: u' = q + v; z_T = solve(rho sum K^T (v - u')); u' += z_T - v
        upf = [ops.lincomb([(1.0, q_fin[i]), (1.0, v[i])]) for i in range(n)]
        for i in range(n):
            terms[i].u, terms[i].u_out, terms[i].v = upf[i].data_ptr(), scratch.data_ptr(), v[i].data_ptr()
        rhs = torch.empty_like(x)
        ops.admm_rhs(rhs, None, rho_tab[T - 1], terms, n)
        ops.fourier_solve(rhs, t0, t1, c0, c1, rho_tab[T - 1], ls_eps(ls), out=x, spec_add=FK)
        u_out = [ops.lincomb([(-1.0, upf[i]), (1.0, v[i]), (-1.0, x)]) for i in range(n)]          # u = -(u' - v + z)
        var.value = x
        s.Kall.update_vars([x])
        return x, v, u_out

    def run_stencil(self, state, rhos, lams, max_iter, method, pbar=False, callback=None):
        """LinearizedADMM (``method='ladmm'``, admm.py:78-100) and PockChambolle (``'pc'``, pc.py:6-40) on fused stages: their
        right-hand sides nest the Psi operators twice (b_i = x - K_i^T(...), then K_i^T b_i), which ``dpx_split_rhs`` evaluates as one
        radius-2 gather pass; the x-update is the same Fourier solve with the fp64 data spectrum added in the Fourier domain, the
        z / dual stage is ``dpx_admm_zupdate`` (LADMM) or ``dpx_pc_dual`` (PC).  Closed-form proxes only; 3-4 passes per iteration
        instead of 10-15."""
        s = self.solver
        ls = s.least_square
        psi = list(s.psi_fns)
        x0 = state[0]
        B, C, H, W = x0.shape
        dev = x0.device
        T = max_iter
        s.Kall.update_vars([x0])
        if T <= 0:
            return state
        rho_tab = schedule_table(rhos, T, B, dev)
        lam_tab = [schedule_table(lams[fn], T, B, dev) for fn in psi]
        FK = self._data_spectrum(x0)
        (t0, c0), (t1, c1) = ls.diag_tables(x0.shape, dev, True)
        n = len(psi)
        var = s.Kall.variables[0]
        rhs = torch.empty_like(x0)
        if method == "ladmm":
            _, v, u = state
            x = x0.clone()
            v, u = [t.contiguous() for t in v], [t.contiguous() for t in u]
            specs = [dict(linop=lc, prox=pc, alpha=float(fn.alpha), lam=lam_tab[i][0], v=v[i], u=u[i]) for i, (fn, (lc, pc)) in enumerate(zip(psi, self.codes))]
            terms = ops.make_terms(specs)
            for it in tqdm(range(T), disable=not pbar):
                for i in range(n):
                    terms[i].lam = lam_tab[i][it].data_ptr()
                ops.split_rhs(rhs, None, x, rho_tab[it], terms, n, 1)
                ops.fourier_solve(rhs, t0, t1, c0, c1, rho_tab[it], ls_eps(ls), out=x, spec_add=FK)
                ops.admm_zupdate(x, terms, n)
                var.value = x
                if callback is not None:
                    s._notify_all_op_current_step(it)
                    callback(iter=it, state=(x, v, u), rho=rhos[..., it], lam={k: val[..., it] for k, val in lams.items()})
            s.Kall.update_vars([x])
            return x, v, u
        # Pock-Chambolle: state (x, [z_i], xbar)
        _, z, xbar = state
        x, xn = x0.clone(), torch.empty_like(x0)
        xbar = xbar.clone()
        z = [t.contiguous() for t in z]
        specs = [dict(linop=lc, prox=pc, alpha=float(fn.alpha), lam=lam_tab[i][0], v=z[i], u=z[i]) for i, (fn, (lc, pc)) in enumerate(zip(psi, self.codes))]
        terms = ops.make_terms(specs)
        for it in tqdm(range(T), disable=not pbar):
            for i in range(n):
                terms[i].lam = lam_tab[i][it].data_ptr()
            ops.pc_dual(xbar, terms, n)                                  # z += r K xbar ; z -= r prox(z, r)
            ops.split_rhs(rhs, None, x, rho_tab[it], terms, n, 0)        # rho sum K^T (x - K^T z)
            ops.fourier_solve(rhs, t0, t1, c0, c1, rho_tab[it], ls_eps(ls), out=xn, spec_add=FK)
            ops.lincomb([(2.0, xn), (-1.0, x)], out=xbar)
            x, xn = xn, x
            var.value = x
            if callback is not None:
                s._notify_all_op_current_step(it)
                callback(iter=it, state=(x, z, xbar), rho=rhos[..., it], lam={k: val[..., it] for k, val in lams.items()})
        s.Kall.update_vars([x])
        return x, z, xbar

    def _data_spectrum(self, x0, chains=1):
        """F(sum_Omega K^T b) in fp64, cached on the solver while the offsets and operator tables are unchanged.
        chains > 1: a list, one packed spectrum per sub-batch chain (every chain's buffer has its own [planes][H][W/2] + Nyquist layout)"""
        s = self.solver
        dev = x0.device
        offs = [fn.offset for fn in s.omega_fns]
        fk_key = (tuple(x0.shape), str(dev), chains) + tuple((id(o), o._version) if o is not None else None for o in offs) + \
            tuple(fn.linop.tables_version() for fn in s.omega_fns)
        cache = getattr(s, "_fk_cache", None)                  # {key: (spectrum, offsets)}: the newest two (one per chain layout in use)
        if cache is None:
            cache = s._fk_cache = {}
        if fk_key in cache:
            return cache[fk_key][0]
        B = int(x0.shape[0])
        parts = []
        for c in range(chains):
            b0, b1 = chain_bounds(B, chains, c)
            FK = None
            for fn, off in zip(s.omega_fns, offs):
                if off is None:
                    continue
                off = off.expand_as(x0) if off.shape != x0.shape else off
                off = off[b0:b1].contiguous()
                cv = _omega_conv(fn)
                otf = cv._tables(x0.shape, dev) if cv is not None else None
                FK = ops.data_spectrum(off, otf, conj=True, out=FK, accumulate=FK is not None)
            parts.append(FK)
        out = parts[0] if chains == 1 else parts
        while len(cache) >= 2:
            cache.pop(next(iter(cache)))
        cache[fk_key] = (out, offs)                            # (the offsets are kept alive: their ids are part of the key)
        return out

    @staticmethod
    def _offset_autograd(fn, x0):
        """The offset b of a recognised Omega term as an autograd-connected tensor when one of its constants requires
        grad (offset = -sum of the constant leaves of  K x + c_1 + ...), else None.  Only routes gradients: the forward
        pass uses the data spectrum built from the native offset."""
        if not torch.is_grad_enabled():
            return None
        b = getattr(fn, "_b", None)                        # sum_squares(linop, b): the observation given as the second argument
        if b is not None:
            val = fn.unwrap(b)
            return val.to(x0.device).expand_as(x0) if isinstance(val, torch.Tensor) and val.requires_grad else None
        consts = [c for c in fn.linop.constants if isinstance(c._value, torch.Tensor) and c._value.requires_grad]
        if not consts:
            return None
        from ..linop import scale as lin_scale

        def walk(node, coef):                              # constant leaves with the factor their scale() chain gives them
            if isinstance(node, Constant):
                return [(coef, node)]
            if type(node) is lin_scale:
                return walk(node.input_nodes[0], coef * float(node.scalar))
            if isinstance(node, lin_sum):
                out = []
                for k in node.input_nodes:
                    if len(k.variables) == 0:
                        sub = walk(k, coef)
                        if sub is None:
                            return None
                        out += sub
                return out
            return None
        leaves = walk(fn.linop, 1.0)
        if leaves is None:
            return None
        tot = None
        for coef, c in leaves:
            val = c._value.to(x0.device) * coef if coef != 1.0 else c._value.to(x0.device)
            tot = val if tot is None else tot + val
        return (-tot).expand_as(x0)

    def _seed_chains(self, x0, v, u, rho_tab, fresh, chains, dev):
        """Sub-batch chains, first half: per chain its spectrum buffers, terms (views of the state) and stream, and the seed passes
        launched -- before the rest of the host-side preparation, which then runs while the GPU is busy (as the one-chain path does).
        Streams are raw handles and forks / joins C calls: torch's stream context managers and wait_stream cost ~15 us each."""
        B, C, H, W = x0.shape
        psi = list(self.solver.psi_fns)
        handles = chain_stream_handles(dev, chains)
        L = be.lib()
        key = (B, C, H, W, chains)
        sizes = _chain_spec_bytes.get(key)
        if sizes is None:
            sizes = _chain_spec_bytes[key] = [L.query("dpx_spectrum_bytes", (chain_bounds(B, chains, c)[1] - chain_bounds(B, chains, c)[0]) * C, H, W) // 2
                                             for c in range(chains)]
        pad = [(sz + 255) // 256 * 256 for sz in sizes]
        pool = ops._bytes(2 * sum(pad), dev)                    # the 2 x chains spectrum buffers in one allocation
        # (addresses by arithmetic: a tensor slice costs ~3 us of host time, and this runs in front of the first launch)
        n = len(psi)
        img = C * H * W * 4                                    # bytes per image of the fp32 state
        pbase, x0p = pool.data_ptr(), x0.data_ptr()
        vps, ups = [t.data_ptr() for t in v], [t.data_ptr() for t in u]
        work, off = [], 0
        for c in range(chains):
            b0, b1 = chain_bounds(B, chains, c)
            terms = (ops.Term * n)()
            for i, (fn, (lc, pc)) in enumerate(zip(psi, self.codes)):
                tm = terms[i]
                tm.linop, tm.prox, tm.alpha = lc, pc, float(fn.alpha)
                tm.v, tm.u = vps[i] + b0 * img, ups[i] + b0 * img
            work.append(dict(b0=b0, b1=b1, shape=(b1 - b0, C, H, W), terms=terms, rho=_chain_table(rho_tab, b0, b1),
                             SA=pbase + off, SB=pbase + off + pad[c], stream=handles[c]))
            off += 2 * pad[c]
        pre = dict(work=work, handles=handles, pool=pool)
        self._pending_chains = pre                             # (run() joins the streams if anything below or behind raises)
        table = ops.ptr(ops.fft_table(H, W, dev))
        L.call("dpx_admm_iter_share", chains)
        try:
            ops.stream_fork(handles[0], handles)                 # (the chains read x0 / the state: produced on the caller's stream)
            for wk in work:
                st = None if wk["stream"] is None else ctypes.c_void_p(wk["stream"])
                if fresh:
                    L.call("dpx_admm_seed_rows_fresh", ctypes.c_void_p(wk["SA"]), ops.ptr(wk["rho"]), ctypes.c_void_p(x0p + wk["b0"] * img), wk["terms"], n,
                           wk["b1"] - wk["b0"], C, H, W, table, st)
                else:
                    L.call("dpx_admm_seed_rows", ctypes.c_void_p(wk["SA"]), ops.ptr(wk["rho"]), wk["terms"], n, wk["b1"] - wk["b0"], C, H, W, table, st)
        finally:
            L.call("dpx_admm_iter_share", 1)
        return pre

    def _run_chains(self, x0, dev, T, n, v, u, x, FK, diag, rho_tab, lam_tab, dual, fresh, chains, pre):
        """The two-kernel iteration as `chains` independent sub-batch chains on separate HIP streams.  The iteration acts per image
        (admm.py:49-59; every table is per channel), so the sub-batches never meet: without a common kernel boundary one chain's column
        pass (load - transform - store in step across its workgroups) runs beside another chain's streaming row pass and the memory
        system stays busy through both kernels' ramps and tails -- 8x3x1024^2: 0.182 -> 0.169 ms per iteration, bit-identical results.
        Every chain has its own spectrum buffers and data spectrum; state, tables and schedules are sub-batch views.  pre: what
        _seed_chains prepared (the seed passes are already running)."""
        s = self.solver
        B, C, H, W = x0.shape
        t0, c0, t1, c1 = diag
        dd = ops.denominator(t0, c0, t1, c1, C, H, W, dev)
        var = s.Kall.variables[0]
        eps = ls_eps(s.least_square)
        x_only = bool(getattr(s, "_x_only", False))
        if dual:
            u_cur, u_nxt = list(u), [torch.empty_like(t) for t in u]
        else:
            u_cur, u_nxt = list(u), [torch.zeros_like(u[0])] * n
        _tr("chains: dd, u_nxt")
        work, handles = pre["work"], pre["handles"]
        img = C * H * W * 4
        vps, ups, uns, xp = [t.data_ptr() for t in v], [t.data_ptr() for t in u_cur], [t.data_ptr() for t in u_nxt], x.data_ptr()
        flags = (0 if dual else be.TERM_NO_DUAL) | (be.TERM_U_ZERO if fresh else 0)
        for wk in work:
            b0, b1 = wk["b0"], wk["b1"]
            for i in range(n):
                tm = wk["terms"][i]
                tm.v, tm.u, tm.u_out = vps[i] + b0 * img, ups[i] + b0 * img, uns[i] + b0 * img
                tm.reserved = flags
            wk["lam"] = [_chain_table(lt, b0, b1) for lt in lam_tab]
            wk["x_out"] = xp + b0 * img
        L = be.lib()
        L.call("dpx_admm_iter_share", chains)
        try:
            # what the caller's stream produced since the seeds were launched (data spectrum, denominators, table slices) is input of every chain
            _tr("chains: terms")
            ops.stream_fork(handles[0], handles)
            # one C call issues every iteration of every chain, chain by chain within an iteration
            par = ops.admm_run_chains([dict(spec_a=wk["SA"], spec_b=wk["SB"], spec_add=fk, terms=wk["terms"], rho_tab=wk["rho"], lam_tabs=wk["lam"],
                                            x_out=wk["x_out"], B=wk["b1"] - wk["b0"], stream=wk["stream"]) for wk, fk in zip(work, FK)],
                                      dd, n, eps, 0, T, T, 2 if x_only else 1, x0.shape, dev)
            _tr("chains: loop issued")
            ops.stream_join(handles[0], handles)                 # (every buffer of the chains stays alive until here: pre["pool"], u_nxt)
        finally:
            L.call("dpx_admm_iter_share", 1)
        if par:
            u_cur, u_nxt = u_nxt, u_cur
        var.value = x
        s.Kall.update_vars([x])
        return (x, v, u_cur) if dual else (x, v)

    def _run_two_kernel(self, shape, dev, T, terms, n, v, u, x, rhs, FK, diag, rho_tab, lam_tab, rhos, lams, pbar, callback, dual=True, seeded=None,
                        fresh=False):
        """power-of-two planes: cols -> rows, two kernels per iteration; x / v only leave the chip on request.
        dual=False (half-quadratic splitting): u holds one shared all-zero buffer per term; the kernels' dual output goes to one
        shared scratch buffer and is ignored when it comes back as input (DPX_TERM_NO_DUAL)"""
        s = self.solver
        B, C, H, W = shape
        t0, c0, t1, c1 = diag
        dd = ops.denominator(t0, c0, t1, c1, C, H, W, dev)
        SA = seeded if seeded is not None else ops.spectrum_buffer(B * C, H, W, dev)     # (run() launches the seed pass first)
        SB = ops.spectrum_buffer(B * C, H, W, dev)
        if dual:
            u_cur, u_nxt = list(u), [torch.empty_like(t) for t in u]
        else:
            u_cur, u_nxt = list(u), [torch.zeros_like(u[0])] * n
        var = s.Kall.variables[0]
        if T == 0:
            return (var.value, v, u) if dual else (var.value, v)
        # seed: row transform of the first right-hand-side increment rho_0 * sum K_i^T (v_i - u_i)
        for i in range(n):
            terms[i].lam = lam_tab[i][0].data_ptr()
        if seeded is None:
            ops.admm_seed_rows(SA, rho_tab[0], terms, n, shape, dev)
        for i in range(n):
            terms[i].u, terms[i].u_out, terms[i].v = u_cur[i].data_ptr(), u_nxt[i].data_ptr(), v[i].data_ptr()
        eps = ls_eps(s.least_square)
        if fresh:                                                # u_i = 0: the first iteration does not stream the duals (DPX_TERM_U_ZERO)
            for i in range(n):
                terms[i].reserved |= be.TERM_U_ZERO
        if callback is None and not pbar:
            # (solve() hands back x alone: the last pass then stores nothing else -- no v, no final dual update: emit mode 2)
            par = ops.admm_run(SA, SB, FK, dd, terms, n, rho_tab, lam_tab, eps, 0, T, T, x, 2 if getattr(s, "_x_only", False) else 1, shape, dev)
            if par:
                u_cur, u_nxt = u_nxt, u_cur
        else:
            for it in tqdm(range(T), disable=not pbar):
                emit = callback is not None or it == T - 1
                par = ops.admm_run(SA, SB, FK, dd, terms, n, rho_tab, lam_tab, eps, it, 1, T, x, emit, shape, dev)
                for i in range(n):
                    terms[i].reserved &= ~be.TERM_U_ZERO
                if par:
                    u_cur, u_nxt = u_nxt, u_cur
                    for i in range(n):
                        terms[i].u, terms[i].u_out = u_cur[i].data_ptr(), u_nxt[i].data_ptr()
                if emit:
                    var.value = x
                if callback is not None:
                    s._notify_all_op_current_step(it)
                    callback(iter=it, state=(x, v, u_cur) if dual else (x, v), rho=rhos[..., it], lam={k: val[..., it] for k, val in lams.items()})
        s.Kall.update_vars([x])
        return (x, v, u_cur) if dual else (x, v)


def ls_eps(ls):
    return 1e-7
