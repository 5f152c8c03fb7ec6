"""``linear_solve`` with an analytic (implicit-function) backward pass
(reference dprox/linalg/custom.py:9-87): dx/db = A^-T, parameter gradients through one extra
application of A at the solution."""
from dataclasses import dataclass, field
from functools import partial

import torch

from .solve import SOLVERS


@dataclass
class LinearSolveConfig:
    rtol: float = 1e-6
    max_iters: int = 100
    verbose: bool = False
    solver_type: str = "cg"
    solver_kwargs: dict = field(default_factory=dict)
    use_analytic_grad: bool = True


def _build_solver(config: LinearSolveConfig):
    if config.solver_type not in SOLVERS:
        raise KeyError(f"solver_type {config.solver_type!r} is not available in the MI355X backend "
                       f"(have {sorted(SOLVERS)})")
    return partial(SOLVERS[config.solver_type], rtol=config.rtol, max_iters=config.max_iters,
                   verbose=config.verbose, **config.solver_kwargs)


def _trainable_parameters(module):
    if not isinstance(module, torch.nn.Module):
        return []
    return [p for p in module.parameters() if p.requires_grad]


class LinearSolve(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, b, config, *Aparams):
        ctx.A = A
        ctx.linear_solver = _build_solver(config)
        x = ctx.linear_solver(A, b)
        ctx.save_for_backward(x, *Aparams)
        return x

    @staticmethod
    def backward(ctx, grad_x):
        grad_B = ctx.linear_solver(ctx.A.T, grad_x.contiguous())
        x = ctx.saved_tensors[0].detach().clone()
        params = _trainable_parameters(ctx.A)
        grads = ()
        if params:
            A = ctx.A.clone()
            with torch.enable_grad():
                loss = -A(x)
            grads = torch.autograd.grad((loss,), _trainable_parameters(A), grad_outputs=(grad_B,),
                                        create_graph=torch.is_grad_enabled(), allow_unused=True)
        return (None, grad_B, None, *grads)


def linear_solve(A, b: torch.Tensor, config: LinearSolveConfig = LinearSolveConfig()):
    """Solve A x = b; ``A(x)`` applies the operator, ``A.T`` / ``A.clone()`` are used by the backward."""
    needs_grad = torch.is_grad_enabled() and (b.requires_grad or bool(_trainable_parameters(A)))
    if config.use_analytic_grad and needs_grad:
        return LinearSolve.apply(A, b, config, *_trainable_parameters(A))
    return _build_solver(config)(A, b)
