"""Quadratic terms and the x-update solver (reference dprox/proxfn/sum_square.py:12-197).

``least_squares`` solves   min_x  sum_Omega ||K x - b||^2 + rho * sum_Psi ||K_i x - b_i||^2 [+ rho ||x - v||^2]

* every linop diagonal in frequency (conv / grad / Variable / scale): ONE pass of hand-written
  kernels -- rFFT2 -> (F + eps) / (sum|OTF|^2 + rho sum|OTF_i|^2 + eps) -> irFFT2 (``dpx_fourier_solve``);
  the denominators are accumulated once as fp64-accurate half-spectrum tables, and the constant
  part of the right-hand side (sum_Omega K^T b) is computed once instead of every iteration;
* every linop diagonal in space (Variable / scale): a fused per-image scaling;
* otherwise conjugate gradients on the HIP primitives (``dprox.linalg``).
"""
from typing import List

import numpy as np
import torch
import torch.nn as nn

from .. import _backend as be
from .. import _ops as ops
from ..linalg import LinearSolveConfig, linear_solve
from ..linop import BlackBox, LinOp, Variable, adjoint, conv, conv_doe, eval, scale, vstack
from ..linop import sum as lin_sum
from .core import ProxFn


def _bytes_hash(a):
    """64-bit fingerprint of an array's bytes: xxh3 (~10 GB/s; the `fast-hash` extra of setup.py) or, where xxhash is missing, blake2b
    with an 8-byte digest (~1 GB/s -- still 64 bits: a 32-bit checksum would keep stale data on a collision).  Cost: one pass over the
    observation per outermost solve (a 100 MB batch: ~10 ms with xxh3, ~0.1 s with blake2b); hand the observation over as a torch
    tensor (edits bump its version counter, nothing is hashed) or call `sum_squares.set_b()` to skip it."""
    mv = memoryview(a).cast("B") if a.size else b""
    try:
        import xxhash
        return xxhash.xxh3_64_intdigest(mv)
    except ImportError:
        import hashlib
        return int.from_bytes(hashlib.blake2b(mv, digest_size=8).digest(), "little")


class sum_squares(ProxFn):
    """||K x - b||_2^2"""
    hip_kind = be.PROX_SUMSQ

    def __init__(self, linop, b=None, eps=1e-7):
        super().__init__(linop)
        self.eps = eps
        # A NumPy observation is converted once -- re-reading it on every use, as the reference does (proxfn/base.py unwrap), rebuilt
        # every cache -- but its in-place edits must not go unnoticed: a tensor's edits bump a version counter, an array's cannot be
        # watched, so the array is kept and its bytes are fingerprinted once per solve (and on every use outside of a solve).
        self._b_np, self._b_fp, self._b_fp_epoch = None, None, None
        if isinstance(b, np.ndarray):
            self._b_np = b
            b = torch.from_numpy(np.ascontiguousarray(b))
        self._b = b

    def set_b(self, b):
        """Replace the observation explicitly (array or tensor): every dependent cache (offset, K^T b, data spectrum) follows -- the call itself
        invalidates them (a counter in the cache key), whatever `id(b)` and version counters say.  A NumPy array's bytes are fingerprinted lazily, at
        its next use.  An in-place edit of a NumPy observation made INSIDE a solve (e.g. by a callback) is only seen at the next solve -- the bytes
        are looked at once per outermost solve; call this to force it."""
        self._b_np, self._b_fp, self._b_fp_epoch = None, None, None
        if isinstance(b, np.ndarray):
            self._b_np = b
            b = torch.from_numpy(np.ascontiguousarray(b))
        self._b = b
        self._set_b_count = getattr(self, "_set_b_count", 0) + 1

    def _np_fingerprint(self):
        ep = be.solve_epoch()
        if ep is None or ep != self._b_fp_epoch or self._b_fp is None:
            a = np.ascontiguousarray(self._b_np)
            fp = (a.shape, str(a.dtype), _bytes_hash(a))
            if fp != self._b_fp:                                        # first look, or edited since the last one: take the contents
                self._b = torch.from_numpy(a)
            self._b_fp, self._b_fp_epoch = fp, ep
        return self._b_fp

    def _offset_key(self):
        k = super()._offset_key() + (getattr(self, "_set_b_count", 0),)
        if self._b is None:
            return k
        # a tensor's in-place edits bump its version counter; anything else (a numpy array) cannot be watched: a key that never
        # compares equal makes every use re-read it, as the reference does
        if self._b_np is not None:
            return k + ((id(self._b_np), self._np_fingerprint()),)
        if isinstance(self._b, torch.Tensor):
            return k + ((id(self._b), self._b._version),)
        ver = getattr(self._b, "_version", None)           # a Placeholder counts its assignments (linop/leaf.py)
        val = getattr(self._b, "_value", None)
        if isinstance(ver, int):
            return k + ((id(self._b), ver, id(val), getattr(val, "_version", None)),)
        return k + (object(),)

    def _compute_offset(self):
        if self._b is not None:
            return self.unwrap(self._b)
        return super()._compute_offset()

    def _prox(self, v, lam):
        return ops.prox(be.PROX_SUMSQ, v, lam, 1.0, None)

    def grad(self, x):
        """K^T (K x - b)"""
        parts = self.grad_parts(x)
        if parts is not None:
            gram_x, ktb = parts
            return gram_x if ktb is None else ops.lincomb([(1.0, gram_x), (-1.0, ktb)])
        r = eval(self.linop, x)
        off = self.offset
        if off is not None:
            r = ops.lincomb([(1.0, r), (-1.0, off.expand_as(r).contiguous())])
        return adjoint(self.linop, r)

    def gram_tables(self, x):
        """(|OTF|^2 table, K^T b) when K is a circular convolution of the variable (cached while the kernel and b are unchanged);
        None otherwise"""
        cv = self.linop
        if isinstance(cv, lin_sum):
            lin = [k for k in cv.input_nodes if k.variables]
            cv = lin[0] if len(lin) == 1 else None
        if type(cv) is not conv or not isinstance(cv.input_nodes[0], Variable) or x.ndim != 4 or x.dtype != torch.float32:
            return None
        if torch.is_grad_enabled() and x.requires_grad:
            return None
        key = (tuple(x.shape), str(x.device), cv.tables_version(), self._offset_key())
        cache = getattr(self, "_gram_cache", None)
        if cache is None or cache[0] != key:
            _, C, H, W = x.shape
            d = cv.get_diag(x, freq=True)                                     # |OTF|^2 on the full grid, [1,C,H,W]
            gram = ops.otf_from_full(torch.complex(d.float(), torch.zeros_like(d, dtype=torch.float32)), C, H, W)
            off = self.offset
            ktb = None if off is None else cv.adjoint(off.expand_as(x).contiguous())
            cache = (key, gram, ktb)
            self._gram_cache = cache
        return cache[1], cache[2]

    def grad_parts(self, x):
        """(K^T K x, K^T b) when K is a circular convolution of the variable: the Gram operator is ONE Fourier multiply by
        |OTF|^2 (3 kernels instead of the 6 of forward + adjoint) and K^T b is constant while b is; None otherwise."""
        tables = self.gram_tables(x)
        if tables is None:
            return None
        return ops.fft_conv(x.contiguous(), tables[0], conj=False), tables[1]


class ext_sum_squares(sum_squares):
    """a data term that brings its own closed-form x-update (``_prox(xtilde, rho, n)``)"""

    def __init__(self, linop, eps=1e-7):
        super().__init__(linop, eps=eps)

    def setup(self, b):
        self.quad_b = b
        return self

    def solve(self, b, rho, eps=1e-6):
        if any(t.is_complex() for t in b) and not all(t.is_complex() for t in b):
            xtilde = ops.clincomb([(1.0, t) for t in b], out_complex=True)
        else:
            xtilde = ops.lincomb([(1.0, t) for t in b]) if len(b) > 1 else b[0]
        return self._prox(xtilde, rho, len(b))


class weighted_sum_squares(sum_squares):
    """||W x - b||^2 with a diagonal weight operator W (reference sum_square.py:51-83): the prox is
    (W^T b + lam v) / (diag(W) + lam), evaluated by ``dpx_wss_prox`` (spatially diagonal weights; the reference's frequency
    branch transforms over the batch / channel axes and is not reproduced)."""

    def __init__(self, linop, weight, b, eps=0):
        super().__init__(linop, b, eps)
        self.weight = weight
        if not self.weight.is_diag():
            raise ValueError("weight {} must be diagonalizable".format(weight))

    @property
    def Ktb(self):
        return adjoint(self.weight, self.unwrap(self._b).to(self.weight.device).contiguous())

    def prox(self, v, lam):
        Ktb = self.Ktb
        diag = self.weight.get_diag(Ktb)
        return ops.wss_prox(v.contiguous(), Ktb, diag, lam)

    def _prox(self, v, lam):
        return self.prox(v, lam)


# ------------------------------------------------------------------------------------------------
# Gram diagonals:  (opaque half-spectrum table | None, constant)
# ------------------------------------------------------------------------------------------------
def _gram_diag(linop, shape, device, freq):
    if isinstance(linop, Variable):
        return None, 1.0
    if isinstance(linop, (conv, conv_doe)):
        if not freq:
            raise ValueError("conv is only diagonal in the frequency domain")
        _, C, H, W = shape
        return linop.accumulate_diag(ops.new_diag(C, H, W, device), 1.0, C, H, W), 0.0
    if isinstance(linop, lin_sum):
        return _gram_diag(linop.input_nodes[0], shape, device, freq)          # sum.py:41-59
    if isinstance(linop, scale):
        t, c = _gram_diag(linop.input_nodes[0], shape, device, freq)          # scale.py:48-61: (s d)(s d)^*
        s = float(linop.scalar)
        if t is None:
            return None, (s * c) ** 2
        return ((t + c) * s) ** 2, 0.0
    ref = torch.zeros(shape, device=device)
    d = linop.get_diag(ref, freq)
    if not freq:
        return ("spatial", torch.as_tensor(d).float().to(device)), 0.0
    _, C, H, W = shape
    return ops.diag_from_full(d, C, H, W, device), 0.0


class least_squares(ProxFn):
    def __init__(self, quad_fns: List[ProxFn], other_fns: List[ProxFn], try_diagonalize=True,
                 try_freq_diagonalize=True, fallback_solver="cg", linear_solve_config=LinearSolveConfig()):
        self.quad_fns_list = list(quad_fns)
        self.other_fns_list = list(other_fns)
        stacked = vstack([fn.linop for fn in self.quad_fns_list + self.other_fns_list])
        self.diagonalizable = stacked.is_gram_diag(freq=False) and try_diagonalize
        self.freq_diagonalizable = stacked.is_gram_diag(freq=True) and try_diagonalize and try_freq_diagonalize
        super().__init__(stacked)
        self.quad_fns = nn.ModuleList(self.quad_fns_list)
        self.other_fns = nn.ModuleList(self.other_fns_list)
        self.try_freq_diagonalize = try_freq_diagonalize
        self.try_diagonalize = try_diagonalize
        self.fallback_solver = fallback_solver
        self.linear_solve_config = linear_solve_config
        self._diag_cache = None
        self._ktb_cache = None
        self.cg_iters = []

    def _prox(self, v, lam):
        return self.solve([], lam, v=v)

    # ---- pieces -----------------------------------------------------------------------------------
    def quad_rhs(self):
        """sum over Omega of K^T offset -- constant while the offsets are (sum_square.py:126-132)"""
        key = tuple(fn._offset_key() for fn in self.quad_fns)
        if self._ktb_cache is None or self._ktb_cache[0] != key:
            parts = []
            for fn in self.quad_fns:
                off = fn.offset
                if off is None:
                    continue
                out = fn.dag.adjoint(off)
                if isinstance(out, LinOp.MultOutput):
                    out = out[0]
                if out is not None:
                    parts.append(out)
            ktb = None if not parts else (parts[0] if len(parts) == 1 else ops.lincomb([(1.0, p) for p in parts]))
            self._ktb_cache = (key, ktb)
        return self._ktb_cache[1]

    def rhs(self, b, rho, v=None):
        terms = []
        ktb = self.quad_rhs()
        if ktb is not None:
            terms.append((1.0, ktb))
        for i, fn in enumerate(self.other_fns):
            kt = fn.dag.adjoint(b[i])
            if kt is not None:
                terms.append((rho, kt))
        if v is not None:
            terms.append((rho, v))
        ref = terms[0][1]
        terms = [(c, t if t.shape == ref.shape else t.expand_as(ref).contiguous()) for c, t in terms]
        return ops.lincomb(terms)

    def diag_tables(self, shape, device, freq):
        dyn = any(isinstance(fn.linop, BlackBox) for fn in list(self.quad_fns) + list(self.other_fns))
        key = (tuple(shape[1:]), str(device), freq) + tuple(fn.linop.tables_version() for fn in list(self.quad_fns) + list(self.other_fns))
        if self._diag_cache is not None and self._diag_cache[0] == key and not dyn:
            return self._diag_cache[1]

        def total(fns):
            tab, const = None, 0.0
            for fn in fns:
                t, c = _gram_diag(fn.linop, shape, device, freq)
                const += c
                if t is not None:
                    tab = t if tab is None else (("spatial", tab[1] + t[1]) if isinstance(t, tuple) else tab + t)
            return tab, const
        out = (total(self.quad_fns), total(self.other_fns))
        self._diag_cache = (key, out)
        return out

    # ---- solvers ----------------------------------------------------------------------------------
    def solve(self, b, rho, v=None, eps=1e-7):
        if self.diagonalizable or self.freq_diagonalizable:
            return self.solve_direct(b, rho, v, eps)
        return self.solve_cg(b, rho, v, self.linear_solve_config)

    def solve_direct(self, b, rho, v=None, eps=1e-7):
        Ktb = self.rhs(b, rho, v)
        B = Ktb.shape[0]
        (t0, c0), (t1, c1) = self.diag_tables(Ktb.shape, Ktb.device, self.freq_diagonalizable)
        if v is not None:
            c1 += 1.0
        rho_v = ops.as_batch_vec(rho, B, Ktb.device)
        if t0 is None and t1 is None:
            # every Gram matrix is a multiple of the identity (Variable / scale chains): the Fourier
            # division degenerates to a per-image scaling (differs from the reference's FFT round trip
            # only by eps/(diag+eps) ~ 1e-8 at pixel 0)
            return ops.lincomb([(1.0 / (c0 + rho_v * c1 + eps), Ktb)])
        if self.freq_diagonalizable:
            return ops.fourier_solve(Ktb, t0, t1, c0, c1, rho_v, eps)
        # user-supplied spatial diagonals (plugin path)
        d = (t0[1] if t0 is not None else 0.0) + c0 + rho_v.view(B, 1, 1, 1) * ((t1[1] if t1 is not None else 0.0) + c1)
        return (Ktb / (d + eps)).float()

    def normal_operator(self, rho, with_identity=False):
        """x -> sum_Omega K^T K x + rho sum_Psi K^T K x (+ rho x): the system matrix of the x-update as an nn.Module with the
        protocol linear_solve expects (callable, .T, .clone()) -- sum_square.py:160-185"""
        quad, other = self.quad_fns, self.other_fns

        class KtK(nn.Module):
            def __init__(self, rho):
                super().__init__()
                self.rho = rho

            def forward(self, x):
                terms = []
                for fn in quad:
                    terms.append((1.0, fn.dag.adjoint(fn.dag.forward(x))))
                for fn in other:
                    terms.append((self.rho, fn.dag.adjoint(fn.dag.forward(x))))
                if with_identity:
                    terms.append((self.rho, x))
                return ops.lincomb([(c, t.contiguous()) for c, t in terms])

            @property
            def T(self):
                return self

            def clone(self):
                return KtK(self.rho)

        return KtK(rho)

    def _masked_fft_system(self, with_identity):
        """(mask, number of rho * I terms) when the normal operator is  A^H A + n rho I  with A = masked_fft(x) -- one subsampled
        Fourier data term and Psi terms acting on x itself (config 4) -- else None"""
        from ..linop import sum as lin_sum
        if len(self.quad_fns) != 1:
            return None
        op = self.quad_fns[0].linop
        if isinstance(op, lin_sum):
            lin = [k for k in op.input_nodes if not getattr(k, "is_constant", False) and type(k).__name__ != "Constant"]
            if len(lin) != 1:
                return None
            op = lin[0]
        if not getattr(op, "is_masked_fft", False) or not isinstance(op.input_nodes[0], Variable):
            return None
        if not all(isinstance(fn.linop, Variable) for fn in self.other_fns):
            return None
        return op.mask, float(len(self.other_fns) + (1 if with_identity else 0))

    @staticmethod
    def _masked_fft_fits(mask, Ktb):
        """the fused CG call takes one mask per plane or per image, real, on planes its LDS-resident transforms hold; anything else
        (a broadcastable [1,1,1,W] mask, a huge plane) goes down the generic cg() loop"""
        B, _, H, W = Ktb.shape
        if not isinstance(mask, torch.Tensor) or mask.is_complex() or mask.numel() not in (H * W, B * H * W):
            return False
        if tuple(mask.shape[-2:]) != (H, W):
            return False
        return bool(be.lib().query("dpx_cg_masked_fft_supported", B, H, W))

    def solve_cg_rhs(self, Ktb, rho_v, with_identity=False, linear_solve_config=None):
        """the CG x-update for an already assembled right-hand side ``Ktb``; records the exit iteration in ``cg_iters``"""
        cfg = linear_solve_config or self.linear_solve_config
        plain = cfg.solver_type == "cg" and not cfg.verbose and not (torch.is_grad_enabled() and Ktb.requires_grad)
        sysm = self._masked_fft_system(with_identity) if plain else None
        if sysm is not None and Ktb.ndim == 4 and Ktb.shape[1] == 1 and Ktb.shape[0] <= 64 and not be.host_mode_skip_fast_cg() \
                and self._masked_fft_fits(sysm[0], Ktb):
            x, n = ops.cg_masked_fft(Ktb.contiguous(), sysm[0], rho_v, sysm[1], cfg.rtol, cfg.max_iters)   # one C call per solve
            self.cg_iters.append(n)
            return x
        A = self.normal_operator(rho_v, with_identity)
        if cfg.solver_type == "cg" and not (torch.is_grad_enabled() and Ktb.requires_grad):
            from ..linalg.solve import cg
            x, n = cg(A, Ktb, rtol=cfg.rtol, max_iters=cfg.max_iters, verbose=cfg.verbose, return_iters=True)
            self.cg_iters.append(n)
            return x
        return linear_solve(A, Ktb, config=cfg)

    def solve_cg(self, b, rho, v=None, linear_solve_config=LinearSolveConfig()):
        Ktb = self.rhs(b, rho, v)
        rho_v = ops.as_batch_vec(rho, Ktb.shape[0], Ktb.device)
        return self.solve_cg_rhs(Ktb, rho_v, with_identity=v is not None, linear_solve_config=linear_solve_config)

    def extra_repr(self) -> str:
        return f"diagonalizable: {self.diagonalizable}; freq_diagonalizable: {self.freq_diagonalizable}"
