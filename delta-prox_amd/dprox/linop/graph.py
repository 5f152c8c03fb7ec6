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
        return node.forward(*ins)

    def forward(self, *values, return_list=False):
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
        ins = node.adjoint(*y) if isinstance(y, LinOp.MultOutput) else node.adjoint(y)
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
