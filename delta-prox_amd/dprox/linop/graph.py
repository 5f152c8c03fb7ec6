"""``CompGraph`` -- evaluates a LinOp expression (forward) and its adjoint.

Same interface as the reference's DAG walker (dprox/linop/comp_graph.py:16-386: ``forward`` /
``adjoint`` with ``return_list``, ``update_vars``, ``sanity_check``, module-level ``eval`` /
``adjoint`` / ``gram`` / ``validate``) but implemented as a plain recursive evaluation of the
expression tree: variables are looked up in an environment, adjoint contributions to a variable
that is used several times are accumulated with one fused HIP pass, and constants that are
"zeroed out" are skipped instead of being materialised as zero tensors.  It is re-entrant (no
module-global result slot).
"""
import torch

from .. import _ops as ops
from .arith import sum as _sum
from .leaf import Constant, Variable
from .node import LinOp


class _LinNode(torch.autograd.Function):
    """A built-in linear node (a fixed linear map evaluated by HIP kernels) as an autograd node: the backward of its forward is
    its adjoint and vice versa.  In the reference these operators are eager torch ops and differentiable by construction
    (tests/test_linop.py:64-80 back-propagates through ``sum``); nodes that bring their own autograd (conv_doe: the PSF) and
    user-defined LinOps (whose forward / adjoint may hold parameters) are called directly."""

    @staticmethod
    def forward(ctx, node, transpose, *args):
        out = (node.adjoint if transpose else node.forward)(*args)
        ctx.node, ctx.transpose, ctx.n_in = node, transpose, len(args)
        ctx.multi = isinstance(out, (list, tuple))
        ctx.shapes = [(a.shape, a.dtype, a.device) for a in (out if ctx.multi else [out])]
        return tuple(out) if ctx.multi else out

    @staticmethod
    def backward(ctx, *gs):
        gs = [torch.zeros(sh, dtype=dt, device=dv) if g is None else g.contiguous() for g, (sh, dt, dv) in zip(gs, ctx.shapes)]
        back = (ctx.node.forward if ctx.transpose else ctx.node.adjoint)(*gs)
        back = list(back) if isinstance(back, (list, tuple)) else [back]
        return (None, None, *back[:ctx.n_in], *([None] * (ctx.n_in - len(back))))


def _native_types():
    from .arith import copy, scale
    from .diagonal import mosaic, mul_color, mul_elementwise
    from .fourier import conv, grad
    return (conv, grad, mosaic, mul_elementwise, mul_color, scale, _sum, copy)


def _apply(node, args, transpose):
    """node.forward(*args) (or adjoint), through _LinNode when something in `args` takes part in autograd"""
    if torch.is_grad_enabled() and type(node) in _native_types() and \
            any(isinstance(a, torch.Tensor) and a.requires_grad for a in args):
        out = _LinNode.apply(node, transpose, *args)
        return LinOp.MultOutput(out) if isinstance(out, tuple) else out
    return (node.adjoint if transpose else node.forward)(*args)


class CompGraph:
    def __init__(self, end, zero_out_constant=False):
        self.end = end
        self.zero_out_constant = zero_out_constant

    @property
    def variables(self):
        return self.end.variables

    # ---- forward ------------------------------------------------------------------------------
    def _fwd(self, node, env):
        if isinstance(node, Variable):
            return env[node.uuid]
        if isinstance(node, Constant):
            return None if self.zero_out_constant else node.value
        if len(node.input_nodes) == 0:           # e.g. vstack([]) when there is no Psi term
            return None
        ins = [self._fwd(ch, env) for ch in node.input_nodes]
        if any(i is None for i in ins):
            if isinstance(node, _sum):
                ins = [i for i in ins if i is not None]
            if not ins or all(i is None for i in ins):
                return None
            if any(i is None for i in ins):      # mixed: materialise the zeros
                ref = next(i for i in ins if i is not None)
                ins = [torch.zeros_like(ref) if i is None else i for i in ins]
        return _apply(node, ins, False)

    def forward(self, *values, return_list=False):
        # (integer images -- uint8 from an image file -- are promoted like torch.fft promotes them in the reference's operators)
        values = [v.float() if isinstance(v, torch.Tensor) and not (v.is_floating_point() or v.is_complex()) else v for v in values]
        values = [v.contiguous() if isinstance(v, torch.Tensor) else v for v in values]
        env = {v.uuid: val for v, val in zip(self.variables, values)}
        y = self._fwd(self.end, env)
        if y is None and values and len(self.end.input_nodes) > 0:
            y = torch.zeros_like(values[0])
        if return_list and y is not None and not isinstance(y, LinOp.MultOutput):
            y = [y]
        return y

    # ---- adjoint ------------------------------------------------------------------------------
    def _adj(self, node, y, acc):
        if y is None or isinstance(node, Constant):
            return
        if isinstance(node, Variable):
            acc[node.uuid] = y if node.uuid not in acc else ops.lincomb([(1.0, acc[node.uuid]), (1.0, y)])
            return
        ins = _apply(node, list(y) if isinstance(y, LinOp.MultOutput) else [y], True)
        kids = list(node.input_nodes)
        if len(kids) == 1:
            self._adj(kids[0], ins, acc)
        else:
            for ch, yi in zip(kids, ins):
                self._adj(ch, yi, acc)

    def adjoint(self, *values, return_list=False):
        acc = {}
        y = LinOp.MultOutput(values) if len(values) > 1 else values[0]
        self._adj(self.end, y, acc)
        vs = self.variables
        outs = [acc.get(v.uuid) for v in vs]
        res = outs[0] if len(outs) == 1 else LinOp.MultOutput(outs)
        if return_list and res is not None and not isinstance(res, LinOp.MultOutput):
            res = [res]
        return res

    # ---- misc ---------------------------------------------------------------------------------
    def update_vars(self, val):
        for i, var in enumerate(self.variables):
            var.value = val[i]

    def x0(self):
        return [torch.zeros(var.shape) for var in self.variables]

    def sanity_check(self, eps=1e-5, shape=(1, 3, 64, 64)):
        """dot-product test <K m, d> == <m, K^T d> (comp_graph.py:342-371)"""
        dev = self.end.device
        m = torch.rand(shape, device=dev)
        d = self.forward(m)
        ds = list(d) if isinstance(d, LinOp.MultOutput) else [d]
        d2 = [torch.rand_like(e) for e in ds]
        m2 = self.adjoint(*d2)
        sum_d = float(sum(ops.bdot(a, b).sum() for a, b in zip(ds, d2)))
        sum_m = float(ops.bdot(m, m2).sum())
        # measured against |Kx| |y| (the reference divides by the inner product itself, which is ill-conditioned
        # for difference operators whose inner products nearly cancel)
        scale = float(sum(ops.bdot(a, a).sum() for a in ds)) ** 0.5 * float(sum(ops.bdot(b, b).sum() for b in d2)) ** 0.5
        rel = abs(sum_m - sum_d) / max(scale, 1e-30)
        print(f"Sanity check {'passed' if rel < eps else 'failed'}, diff={abs(sum_m - sum_d)} rel_diff={rel}")
        return rel < eps

    def __str__(self):
        return self.__class__.__name__


def validate(linop, **kw):
    return CompGraph(linop).sanity_check(**kw)


def eval(linop, *inputs, zero_out_constant=True):
    return CompGraph(linop, zero_out_constant).forward(*inputs)


def adjoint(linop, *inputs, zero_out_constant=True):
    return CompGraph(linop, zero_out_constant).adjoint(*inputs)


def gram(linop, *inputs, zero_out_constant=True):
    K = CompGraph(linop, zero_out_constant)
    out = K.forward(*inputs)
    return K.adjoint(*out) if isinstance(out, LinOp.MultOutput) else K.adjoint(out)
