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
    if len(r.shape) == 1:
        r = r.view(r.shape[0], 1, 1, 1)
    return r


def to_tensor(x, batch=False):
    if isinstance(x, dict):
        return {k: to_tensor(v, batch) for k, v in x.items()}
    return to_torch_tensor(x, batch)


def to_device(x, device):
    if x is None:
        return None
    if isinstance(x, dict):
        return {k: to_device(v, device) for k, v in x.items()}
    if x.is_complex():
        return x.to(device=device, dtype=torch.complex64)
    return x.to(device=device, dtype=torch.float32)       # the backend computes in fp32


def move(*args, device):
    return [to_device(a, device) for a in args]


def auto_convert_to_tensor(names: List[str], batchify: List[str]):
    """converts the *keyword* arguments listed in ``names`` (positional ones are left alone, like the reference)"""
    def outer(fn):
        def wrapper(*args, **kwargs):
            for k, v in kwargs.items():
                if k in names and v is not None:
                    kwargs[k] = to_tensor(v, batch=k in batchify)
            return fn(*args, **kwargs)
        return wrapper
    return outer


def isscalar(x):
    return np.isscalar(x) or (isinstance(x, torch.Tensor) and len(x.shape) == 0)


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
        device = self.device
        if device.type != "cuda" and not be.host_mode():
            raise be.DpxError(f"solver lives on {device}: the MI355X backend has no CPU path; "
                              "compile(..., device='cuda') / Problem.solve(device='cuda')")
        x0, rhos, lams, max_iter = self.defaults(x0, rhos, lams, max_iter)
        x0, rhos, lams = move(x0, rhos, lams, device=device)
        x0 = x0.contiguous()
        state = self.initialize(x0, **kwargs)
        state = self.iters(state, rhos, lams, max_iter, pbar, callback=callback)
        return state if return_full_states else state[0]

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
        if rhos is None:
            rhos = 1.0
        if lams is None:
            lams = 0.02
        if isscalar(rhos):
            rhos = to_tensor([float(rhos)] * max_iter)
        if isscalar(lams):
            lams = {fn: to_tensor([float(lams)] * max_iter) for fn in self.psi_fns}
        lams = {k: to_tensor([float(v)] * max_iter) if isscalar(v) else v for k, v in lams.items()}
        return x0, rhos, lams, max_iter

    # ---- state packing helpers (used by learned step-size policies) -------------------------------
    def pack(self, state):
        flat = []
        for s in state:
            flat += s if isinstance(s, list) else [s]
        return torch.cat(flat, dim=1)

    def unpack(self, tensor):
        parts = list(torch.split(tensor, tensor.shape[1] // self.state_dim, dim=1))
        out, pos = [], 0
        for d in self.state_split:
            if d == 1:
                out.append(parts[pos])
                pos += 1
            else:
                out.append(parts[pos:pos + d[0]])
                pos += d[0]
        return out

    @property
    def state_dim(self):
        n = 0
        for s in self.state_split:
            n += sum(s) if isinstance(s, list) else s
        return n

    @property
    def nparams(self):
        return NotImplementedError

    @property
    def state_split(self):
        return NotImplementedError
