"""User-facing entry points: ``compile``, ``specialize``, ``Problem``
(reference dprox/algo/primitives.py:24-95, dprox/algo/problem.py:13-55, specialization/unroll.py:14-58)."""
import copy
from functools import partial
from typing import List, Union

import torch
import torch.nn as nn

from .. import _backend as be
from ..linalg import LinearSolveConfig
from ..proxfn import ProxFn
from .driver import Algorithm
from .gradient import ProximalGradientDescent
from .splitting import ADMM, HQS, ADMM_vxu, LinearizedADMM, PockChambolle

SOLVERS = {
    "admm": ADMM,
    "admm_vxu": ADMM_vxu,
    "ladmm": LinearizedADMM,
    "hqs": HQS,
    "pc": PockChambolle,
    "pgd": ProximalGradientDescent,
}


def _default_device():
    return "cuda"


def _resolve_device(device):
    device = torch.device(device) if isinstance(device, str) else device
    if device.type != "cuda" and not be.host_mode():
        raise be.DpxError(f"device={device}: this is the MI355X backend of Delta-Prox -- solvers run on HIP devices only "
                          "(no CPU path). Use device='cuda'.")
    if be.host_mode():
        return torch.device("cpu")
    return device


def compile(prox_fns: List[ProxFn], method: str = "admm", device: Union[str, torch.device] = "cuda", **kwargs):
    """Compile an objective (a list / sum of proxable functions) into a proximal solver."""
    if method not in SOLVERS:
        raise KeyError(f"unknown method {method!r}; valid methods are {sorted(SOLVERS)}")
    algorithm = SOLVERS[method]
    if isinstance(prox_fns, ProxFn):
        prox_fns = [prox_fns]
    psi_fns, omega_fns = algorithm.partition(prox_fns)
    solver = algorithm.create(psi_fns, omega_fns, **kwargs)
    return solver.to(_resolve_device(device))


class UnrolledSolver(nn.Module):
    """one (deep-copied) solver per unrolled step, optionally with learnable rho / lambda schedules --
    specialization/unroll.py:21-58"""

    def __init__(self, solver: Algorithm, max_iter, share=False, learned_params=False, dtype="f32"):
        super().__init__()
        _set_unroll_dtype(solver, dtype)
        if not share:
            self.solvers = nn.ModuleList([solver] + [copy.deepcopy(solver) for _ in range(max_iter - 1)])
        else:
            self.solver = solver
            self.solvers = [solver for _ in range(max_iter)]
        self.max_iter, self.share = max_iter, share
        self.learned_params = learned_params
        if learned_params:
            self.rhos = nn.Parameter(torch.ones(max_iter))
            self.lams = {}
            # parameter names follow the reference (setattr(self, str(fn), lam), unroll.py:35-38: the class name of the term), so
            # that its checkpoints load; there a second term of the same class silently takes the name over -- here the earlier
            # one keeps a suffixed name instead of dropping out of parameters() / state_dict()
            names = [str(fn) for fn in solver.psi_fns]
            for i, fn in enumerate(solver.psi_fns):
                lam = nn.Parameter(torch.ones(max_iter))
                last = i == max(j for j, n in enumerate(names) if n == names[i])
                setattr(self, names[i] if last else f"{names[i]}#{i}", lam)
                self.lams[fn] = lam

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """A reference checkpoint holds ``rhos`` and ONE entry per term class (unroll.py:35-38): the suffixed names this class gives to
        earlier terms of a repeated class (``norm1#0``) are optional on load -- absent, they start from the class's un-suffixed entry
        (what the reference trained for the term that kept the name), so ``load_state_dict(strict=True)`` accepts its checkpoints."""
        if self.learned_params:
            for name, _ in list(self.named_parameters(recurse=False)):
                if "#" in name and prefix + name not in state_dict and prefix + name.split("#")[0] in state_dict:
                    state_dict = dict(state_dict) if not isinstance(state_dict, dict) or "#patched" not in state_dict else state_dict
                    state_dict[prefix + name] = state_dict[prefix + name.split("#")[0]]
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def solve(self, x0=None, rhos=None, lams=None, max_iter=None, **kwargs):
        from .driver import to_tensor
        first = self.solvers[0]
        max_iter = self.max_iter if max_iter is None else max_iter
        x0 = to_tensor(x0, batch=True)
        if self.learned_params:
            rhos, lams = self.rhos, self.lams
        else:
            _, rhos, lams, _ = first.defaults(x0, to_tensor(rhos) if rhos is not None else None,
                                              to_tensor(lams) if lams is not None else None, max_iter)
        dev = first.device
        x0 = x0.to(dev).float().contiguous()
        with be.solve_scope("solve"):
            state = first.initialize(x0)
            for it in range(max_iter):
                solver = self.solvers[it]
                rho = rhos[..., it:it + 1].to(dev)
                # schedules are keyed by the FIRST solver's Psi terms; step `it` uses its own clone of each term
                lam = {fn_i: lams[fn_0][..., it:it + 1].to(dev) for fn_0, fn_i in zip(first.psi_fns, solver.psi_fns)}
                solver._notify_all_op_current_step(it)
                state = solver.iters(state, rho, lam, 1, False)
        return state[0]


UNROLL_DTYPES = ("f32", "bf16")
UNROLL_BF16 = True          # specialize(..., method='unroll', dtype='bf16') is available (bench.py looks for this)


def _set_unroll_dtype(solver, dtype):
    """``dtype='bf16'`` (BASELINE config 5): the unrolled iteration still computes in fp32 -- the x-update amplifies round-off by up
    to 1e5, see DESIGN.md -- but the history kept for the backward pass (rhs, x, v_i of every iteration) is stored in bf16: a
    third of the fp32 history's bytes; gradients w.r.t. the lambda schedules and the observation are unaffected (they read
    threshold masks, which survive the rounding), those w.r.t. the rho schedule carry ~1e-3 relative bf16 rounding."""
    if dtype not in UNROLL_DTYPES:
        raise ValueError(f"unroll dtype must be one of {UNROLL_DTYPES}, got {dtype!r}")
    solver.unroll_dtype = dtype


def build_unrolled_solver(solver, share=True, dtype="f32", **kwargs):
    if share:
        _set_unroll_dtype(solver, dtype)
        solver.solve = partial(solver.solve, **kwargs)
        return solver
    return UnrolledSolver(solver, share=share, dtype=dtype, **kwargs)


SPECAILIZATIONS = {"unroll": build_unrolled_solver}


def specialize(solver: Algorithm, method: str = "unroll", device: Union[str, torch.device] = "cuda", **kwargs):
    if method not in SPECAILIZATIONS:
        raise NotImplementedError(f"specialization {method!r} (training strategy) is outside the MI355X hot-path backend; "
                                  f"available: {sorted(SPECAILIZATIONS)}")
    solver = SPECAILIZATIONS[method](solver, **kwargs)
    return solver.to(_resolve_device(device))


class Problem:
    def __init__(self, prox_fns: Union[ProxFn, List[ProxFn]], constraints=[], absorb=True, merge=True,
                 try_diagonalize=True, try_freq_diagonalize=True, linear_solve_config=LinearSolveConfig()):
        if isinstance(prox_fns, ProxFn):
            prox_fns = [prox_fns]
        self.prox_fns = prox_fns
        self.absorb, self.merge = absorb, merge
        self.solver_args = dict(try_diagonalize=try_diagonalize, try_freq_diagonalize=try_freq_diagonalize,
                                linear_solve_config=linear_solve_config)
        self.solver = None

    @property
    def objective(self):
        return self.prox_fns

    def solve(self, method="admm", device="cuda", **kwargs):
        args = self.solver_args if method != "pgd" else {}
        self.solver = compile(self.prox_fns, method=method, device=device, **args)
        return self.solver.solve(**kwargs)
