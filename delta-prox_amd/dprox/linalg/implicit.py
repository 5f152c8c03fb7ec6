"""Linear solves ``A x = b`` with an implicit-function backward pass (reference dprox/linalg/custom.py:9-87).

Differentiating through the iterations of an inner Krylov solver is wasteful and unstable; like the reference, the
backward pass solves one more system with the transposed operator,

    dL/db = A^-T (dL/dx),

and obtains the gradients of the operator's own parameters theta from a single extra application of A at the solution,

    dL/dtheta = -(dL/db)^T (dA/dtheta) x      (vector-Jacobian product of  theta -> -A_theta(x)  with cotangent dL/db).

Protocol of ``A`` (same as the reference's): callable ``A(x)``; ``A.T`` the transposed operator; ``A.clone()`` a copy whose
parameters are the tensors handed to ``LinearSolve.apply`` (needed so that autograd can attribute ``dA/dtheta``).
The solver itself (``dprox.linalg.solve``: CG on the HIP primitives ``dpx_lincomb`` / ``dpx_bdot`` / ``dpx_bgram``) is picked
by ``LinearSolveConfig.solver_type``.
"""
import dataclasses
import functools
from typing import Callable, Sequence

import torch

from .solve import SOLVERS


@dataclasses.dataclass
class LinearSolveConfig:
    """stopping rule and solver selection (custom.py:9-26; defaults are the reference's)"""
    rtol: float = 1e-6
    max_iters: int = 100
    verbose: bool = False
    solver_type: str = "cg"
    solver_kwargs: dict = dataclasses.field(default_factory=dict)
    use_analytic_grad: bool = True


def make_solver(config: LinearSolveConfig) -> Callable:
    """the registry entry bound to the configured tolerances"""
    try:
        fn = SOLVERS[config.solver_type]
    except KeyError:
        raise KeyError(f"solver_type {config.solver_type!r} is not available in the MI355X backend (have {sorted(SOLVERS)})") from None
    return functools.partial(fn, rtol=config.rtol, max_iters=config.max_iters, verbose=config.verbose, **config.solver_kwargs)


_build_solver = make_solver          # reference name


def operator_parameters(A) -> Sequence[torch.Tensor]:
    """the trainable tensors of an operator given as an nn.Module (plain callables have none)"""
    if isinstance(A, torch.nn.Module):
        return [p for p in A.parameters() if p.requires_grad]
    return []


_trainable_parameters = operator_parameters


def _parameter_vjp(A, x, cotangent):
    """-(cotangent)^T dA/dtheta x  for every trainable theta of A, via one differentiable application of a clone"""
    if not operator_parameters(A):
        return ()
    probe = A.clone()
    with torch.enable_grad():
        minus_Ax = -probe(x)
    return torch.autograd.grad((minus_Ax,), operator_parameters(probe), grad_outputs=(cotangent,),
                               create_graph=torch.is_grad_enabled(), allow_unused=True)


class LinearSolve(torch.autograd.Function):
    """forward: x = solver(A, b);  backward: one transposed solve + one operator VJP (see module docstring)"""

    @staticmethod
    def forward(ctx, A, b, config, *theta):
        solver = make_solver(config)
        x = solver(A, b)
        ctx.operator, ctx.solver = A, solver
        ctx.save_for_backward(x, *theta)
        return x

    @staticmethod
    def backward(ctx, grad_x):
        x = ctx.saved_tensors[0].detach().clone()
        grad_b = ctx.solver(ctx.operator.T, grad_x.contiguous())
        return (None, grad_b, None, *_parameter_vjp(ctx.operator, x, grad_b))


def linear_solve(A, b: torch.Tensor, config: LinearSolveConfig = LinearSolveConfig()):
    """Solve ``A x = b``.  With ``config.use_analytic_grad`` (default) and something to differentiate (``b`` or parameters
    of ``A``), the result carries the implicit backward above; otherwise the solver is called directly."""
    theta = operator_parameters(A)
    differentiable = torch.is_grad_enabled() and (b.requires_grad or len(theta) > 0)
    if config.use_analytic_grad and differentiable:
        return LinearSolve.apply(A, b, config, *theta)
    return make_solver(config)(A, b)
