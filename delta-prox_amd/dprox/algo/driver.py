"""``Algorithm`` -- the iteration driver shared by every proximal solver
(reference dprox/algo/base.py:58-275).

Keeps the reference's contract: keyword-only tensor conversion of ``x0 / rhos / lams``
(base.py:20-33), defaults ``rho = 1.0, lam = 0.02, max_iter = 24`` (:205-218), scalar ``lams`` apply to
the Psi terms and are keyed by ProxFn object, ``Variable.value`` is updated around every step (:174-178),
``callback(iter=, state=, rho=, lam=)`` after every step (:154-155), ``return_full_states``.
State tensors live in HBM for the whole solve; per-iteration scalars are uploaded once as [T, B]
device arrays so the hot loop issues kernels only.
"""
import abc
from typing import Callable, Iterable, List, Union

import numpy as np
import torch
import torch.nn as nn
from tqdm import tqdm

from .. import _backend as be
from ..linop import CompGraph, vstack
from ..proxfn import ProxFn
from ..utils import to_torch_tensor


def expand(r):
    """[B] -> [B,1,1,1] (broadcast against NCHW); anything else unchanged"""
    return r.reshape(-1, 1, 1, 1) if r.ndim == 1 else r


def _map_leaves(fn, obj):
    """apply fn to a tensor-like or to every value of a {ProxFn: schedule} dict; None passes through"""
    if obj is None:
        return None
    if isinstance(obj, dict):
        return {key: _map_leaves(fn, val) for key, val in obj.items()}
    return fn(obj)


def to_tensor(x, batch=False):
    return _map_leaves(lambda leaf: to_torch_tensor(leaf, batch), x)


def to_device(x, device):
    """onto the solver's device in the backend's compute types: fp32, or complex64 for complex iterates"""
    def place(t):
        return t.to(device=device, dtype=torch.complex64 if t.is_complex() else torch.float32)
    return _map_leaves(place, x)


def move(*args, device):
    return [to_device(a, device) for a in args]


def auto_convert_to_tensor(names: List[str], batchify: List[str]):
    """Decorator: the *keyword* arguments listed in ``names`` become tensors (those in ``batchify`` also get the NCHW
    treatment of ``to_torch_tensor(batch=True)``).  Positional arguments are deliberately left alone -- the reference
    behaves the same way (base.py:20-33), so ``solve(x0=img)`` and ``solve(img)`` differ for HWC arrays."""
    wanted, batched = frozenset(names), frozenset(batchify)

    def decorate(fn):
        def call(*args, **kwargs):
            converted = {k: (to_tensor(v, batch=k in batched) if (k in wanted and v is not None) else v) for k, v in kwargs.items()}
            return fn(*args, **converted)
        return call
    return decorate


def isscalar(x):
    return np.isscalar(x) or (isinstance(x, torch.Tensor) and x.ndim == 0)


def _constant_schedule(value, steps):
    return to_tensor([float(value)] * steps)


def _restarted_callback(callback):
    """the user's callback for the second run of a solve that was re-run (split-f16 range trap): passes restarted=True when the
    callback's signature takes it, the callback unchanged otherwise"""
    if callback is None:
        return None
    import inspect
    try:
        ps = inspect.signature(callback).parameters
    except (TypeError, ValueError):
        return callback
    if "restarted" in ps or any(p.kind is inspect.Parameter.VAR_KEYWORD for p in ps.values()):
        return lambda **kw: callback(restarted=True, **kw)
    return callback


class Algorithm(nn.Module):
    @classmethod
    @abc.abstractmethod
    def partition(cls, prox_fns: List[ProxFn]):
        return NotImplementedError

    @classmethod
    def create(cls, *args, **kwargs):
        return cls(*args, **kwargs)

    def __init__(self, psi_fns: List[ProxFn], omega_fns: List[ProxFn]):
        super().__init__()
        self.psi_fns = nn.ModuleList(psi_fns)
        self.omega_fns = nn.ModuleList(omega_fns)
        self.K = CompGraph(vstack([fn.linop for fn in psi_fns]))
        self.Kall = CompGraph(vstack([fn.linop for fn in list(psi_fns) + list(omega_fns)]))

    @property
    def device(self):
        return next(self.parameters()).device

    @auto_convert_to_tensor(["x0", "rhos", "lams"], batchify=["x0"])
    def solve(self, x0: Union[torch.Tensor, np.ndarray] = None, rhos: Union[float, Iterable[float]] = None,
              lams: Union[float, Iterable[float], dict] = None, max_iter: int = 24, pbar: bool = False,
              callback: Callable = None, return_full_states=False, **kwargs) -> torch.Tensor:
        """Not thread-safe, like the reference's (module-global state in linop/comp_graph.py:201-202): a solve keeps per-solver flags
        (``_x_only``, the fresh-state marker), the sub-batch chains set a per-process hint in the library (dpx_admm_iter_share) and
        share one set of chain events per (host thread, device).  One solve at a time per process and solver; one process per GPU
        (dprox.distributed) is the supported way to use several GPUs."""
        device = self.device
        if device.type != "cuda" and not be.host_mode():
            raise be.DpxError(f"solver lives on {device}: the MI355X backend has no CPU path; "
                              "compile(..., device='cuda') / Problem.solve(device='cuda')")
        x0, rhos, lams, max_iter = self.defaults(x0, rhos, lams, max_iter)
        # every kernel of the solve is issued on the current stream of the SOLVER's GPU (the C ABI takes a raw stream handle:
        # launching with another device current would run GPU-k pointers on GPU-0's stream)
        def run(callback=callback):
            with be.device_guard(device), be.solve_scope("solve"):
                xs, rs, ls = move(x0, rhos, lams, device=device)
                state = self._initial_state(xs.contiguous(), **kwargs)
                # only x leaves solve(): a solver may skip whatever of its LAST iteration x does not depend on (fused.py: the final
                # z / dual update of the two-kernel iteration); callbacks and return_full_states see complete states
                self._x_only = callback is None and not return_full_states
                try:
                    return self.iters(state, rs, ls, max_iter, pbar, callback=callback)
                finally:
                    self._x_only = False
        try:
            state = run()
        except be.F16RangeError:
            # a split-f16 denoiser layer met an operand outside the binary16 range: that solve is invalid -- run it again from x0 on
            # the split-bf16 arithmetic.  Side effects, stated: the networks keep that mode; the user's callback has already seen the
            # iterations of the abandoned run and sees those of the second run too -- a callback that accepts a `restarted` keyword
            # (or **kwargs) is told so with restarted=True on every call of the second run; the warning points at solve()'s caller.
            nets = [m for fn in list(self.psi_fns) + list(self.omega_fns) if isinstance(getattr(fn, "denoiser", None), torch.nn.Module)
                    for m in fn.denoiser.modules()]
            if not be.f16_fallback(nets, "solve", stacklevel=4):
                raise
            state = run(_restarted_callback(callback))
        return state if return_full_states else state[0]

    def _initial_state(self, x0, **kwargs):
        """solve()'s initial state (= initialize unless a solver knows a cheaper equivalent for its own iteration)"""
        return self.initialize(x0, **kwargs)

    def iters(self, state, rhos, lams, max_iter, pbar=False, callback=None):
        for it in tqdm(range(max_iter), disable=not pbar):
            rho = rhos[..., it]
            lam = {k: v[..., it] for k, v in lams.items()}
            self._notify_all_op_current_step(it)
            state = self.iter(state, rho, lam)
            if callback is not None:
                callback(iter=it, state=state, rho=rho, lam=lam)
        return state

    def iter(self, state, rho, lam):
        self.Kall.update_vars([state[0]])
        state = self._iter(state, rho, lam)
        self.Kall.update_vars([state[0]])
        return state

    @abc.abstractmethod
    def _iter(self, state, rho, lam):
        return NotImplementedError

    def _notify_all_op_current_step(self, step):
        def visit(op):
            op.step = step
            for node in op.input_nodes:
                visit(node)
        for fn in list(self.psi_fns) + list(self.omega_fns):
            fn.step = step
            visit(fn.linop)

    def defaults(self, x0=None, rhos=None, lams=None, max_iter=24):
        """rho = 1.0 and lam = 0.02 when absent; scalars become constant schedules of length max_iter; a scalar ``lams``
        applies to every Psi term, keyed by the ProxFn object (base.py:205-218)"""
        rhos = 1.0 if rhos is None else rhos
        lams = 0.02 if lams is None else lams
        if isscalar(rhos):
            rhos = _constant_schedule(rhos, max_iter)
        if isscalar(lams):
            lams = {fn: lams for fn in self.psi_fns}
        lams = {fn: (_constant_schedule(val, max_iter) if isscalar(val) else val) for fn, val in lams.items()}
        return x0, rhos, lams, max_iter

    # ---- state <-> one [B, n*C, H, W] tensor (learned step-size policies look at the packed state) -----------
    def pack(self, state):
        pieces = []
        for entry in state:
            pieces.extend(entry if isinstance(entry, (list, tuple)) else [entry])
        return torch.cat(pieces, dim=1)

    def unpack(self, tensor):
        chunks = list(torch.split(tensor, tensor.shape[1] // self.state_dim, dim=1))
        state, cursor = [], 0
        for spec in self.state_split:
            if isinstance(spec, (list, tuple)):
                state.append(chunks[cursor:cursor + spec[0]])
                cursor += spec[0]
            else:
                state.append(chunks[cursor])
                cursor += 1
        return state

    @property
    def state_dim(self):
        return sum(sum(spec) if isinstance(spec, (list, tuple)) else spec for spec in self.state_split)

    @property
    def nparams(self):
        return NotImplementedError

    @property
    def state_split(self):
        return NotImplementedError
