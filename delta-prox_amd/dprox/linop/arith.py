"""Structural nodes: ``sum`` / ``copy`` / ``scale`` / ``vstack`` / ``split``
(reference dprox/linop/sum.py:6-109, scale.py:7-80, vstack.py:6-120).  The additions and
scalings run as one fused HIP pass (``dpx_lincomb``)."""
import numpy as np
import torch

from .. import _ops as ops
from .node import LinOp


def _zero_preserving(node):
    """a chain of built-in, shape-preserving linear nodes over one variable (exact types: a user subclass may do anything)"""
    from .fourier import conv, grad
    from .leaf import Variable
    if type(node) is Variable:
        return True
    if type(node) in (conv, grad, scale) and getattr(node, "circular", True):
        return all(_zero_preserving(k) for k in node.input_nodes)
    return False


def _expand_to(t, ref):
    return t if t.shape == ref.shape else t.expand_as(ref).contiguous()


class sum(LinOp):
    """y = sum_i x_i ; adjoint copies y to every input"""

    def forward(self, *inputs, **kwargs):
        ref = inputs[0]
        return ops.lincomb([(1.0, _expand_to(t.to(ref.device), ref)) for t in inputs])

    def adjoint(self, y, **kwargs):
        outs = LinOp.MultOutput([y for _ in self.input_nodes])
        return outs if len(outs) > 1 else outs[0]

    @property
    def offset(self):
        """``K x + c_1 + ...`` with K a chain of the built-in shape-preserving linear nodes (the data term ``conv(x) - b``): the value
        at x = 0 is the sum of the constants -- no transform of an all-zero image (linop/base.py:117-129 evaluates the whole graph;
        a circular convolution / difference / scaling of zeros is exactly zero, so the result is the same tensor)"""
        consts = [k for k in self.input_nodes if len(k.variables) == 0]       # Constant / Placeholder leaves and subtrees over them
        others = [k for k in self.input_nodes if len(k.variables) > 0]
        vs = self.variables
        if consts and others and all(_zero_preserving(k) for k in others) and len(vs) == 1 and vs[0]._value is not None:
            vals = [c.value for c in consts]
            if all(v is not None and v.shape == vs[0]._value.shape for v in vals):
                return vals[0] if len(vals) == 1 else ops.lincomb([(1.0, v) for v in vals])
        return LinOp.offset.fget(self)

    def is_diag(self, freq=False):
        return all(a.is_diag(freq) for a in self.input_nodes)

    def is_gram_diag(self, freq=False):
        return all(a.is_gram_diag(freq) for a in self.input_nodes)

    def get_diag(self, ref, freq=False):
        return self.input_nodes[0].get_diag(ref, freq)        # sum.py:41-59: first input only

    def norm_bound(self, input_mags):
        return float(np.sum(input_mags))


class copy(sum):
    """y_i = x for every consumer ; adjoint sums"""

    def __init__(self, arg):
        super().__init__([arg])

    def forward(self, x, **kwargs):
        return sum.adjoint(self, x)

    def adjoint(self, *ys, **kwargs):
        return sum.forward(self, *ys)

    def norm_bound(self, input_mags):
        return input_mags[0]


class scale(LinOp):
    def __init__(self, scalar, arg):
        assert np.isscalar(scalar)
        self.scalar = scalar
        super().__init__([arg])

    def forward(self, x, **kwargs):
        return ops.lincomb([(float(self.scalar), x)])

    def adjoint(self, y, **kwargs):
        return self.forward(y)

    def is_gram_diag(self, freq=False):
        return self.input_nodes[0].is_gram_diag(freq)

    def is_diag(self, freq=False):
        return self.input_nodes[0].is_diag(freq)

    def get_diag(self, ref, freq=False):
        d = self.input_nodes[0].get_diag(ref, freq) * self.scalar
        return d * torch.conj(d)

    def norm_bound(self, input_mags):
        return abs(self.scalar) * input_mags[0]


class vstack(LinOp):
    """stacks the outputs of several operators acting on the same variable(s)"""

    def forward(self, *inputs, **kwargs):
        return LinOp.MultOutput(inputs) if len(inputs) > 1 else inputs[0]

    def adjoint(self, *inputs, **kwargs):
        return LinOp.MultOutput(inputs) if len(inputs) > 1 else inputs[0]

    def is_gram_diag(self, freq=False):
        return all(a.is_gram_diag(freq) for a in self.input_nodes)

    def norm_bound(self, input_mags):
        return float(np.linalg.norm(input_mags, 2))


class split(vstack):
    def __init__(self, output_nodes):
        self.output_nodes = output_nodes
        super().__init__(output_nodes)

    def forward(self, *inputs, **kwargs):
        return vstack.adjoint(self, *inputs, **kwargs)

    def adjoint(self, *inputs, **kwargs):
        return vstack.forward(self, *inputs, **kwargs)

    def norm_bound(self, input_mags):
        return input_mags[0]
