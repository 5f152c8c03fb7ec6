"""Structural nodes: ``sum`` / ``copy`` / ``scale`` / ``vstack`` / ``split``
(reference dprox/linop/sum.py:6-109, scale.py:7-80, vstack.py:6-120).  The additions and
scalings run as one fused HIP pass (``dpx_lincomb``)."""
import numpy as np
import torch

from .. import _ops as ops
from .node import LinOp


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
