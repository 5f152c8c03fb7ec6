"""``LinOp`` -- the node protocol of the linear-operator expression graph.

Mirrors the reference's plugin surface (dprox/linop/base.py:18-247): subclasses implement
``forward(*inputs)`` / ``adjoint(*outputs)`` and optionally ``is_diag`` / ``is_gram_diag`` /
``get_diag``; ``+ - * /`` build ``sum`` / ``scale`` / ``Constant`` nodes.  Nodes are ``nn.Module``s
so that ``solver.to(device)`` and ``parameters()`` behave like the reference's.
All arithmetic done by the built-in nodes runs through the HIP kernels in ``dprox._ops``.
"""
import copy as _copy
import itertools

import numpy as np
import torch
import torch.nn as nn

from ..utils import to_torch_tensor

_ids = itertools.count()


def cast_to_const(expr):
    from .leaf import Constant
    return expr if isinstance(expr, LinOp) else Constant(expr)


class LinOp(nn.Module):
    class MultOutput(list):
        """several outputs of one node (sum.adjoint, vstack.forward, ...)"""

    def __init__(self, input_nodes=()):
        super().__init__()
        self.input_nodes = nn.ModuleList([cast_to_const(n) for n in input_nodes])
        self.linop_id = next(_ids)
        # device anchor: moves with .to(device) like the reference's `dummy` parameter (base.py:35)
        self.dummy = nn.Parameter(torch.tensor(0.0), requires_grad=False)
        self.step = 0     # current iteration, set by Algorithm before every step (algo/base.py:158-172)

    # ---- to be provided by subclasses ---------------------------------------------------------
    def forward(self, *inputs, **kwargs):
        return NotImplemented

    def adjoint(self, *outputs, **kwargs):
        return NotImplemented

    def is_diag(self, freq=False):
        return False

    def is_gram_diag(self, freq=False):
        return self.is_diag(freq)

    def get_diag(self, ref, freq=False):
        return NotImplemented

    def norm_bound(self, input_mags):
        return NotImplemented

    def tables_version(self):
        """changes whenever a value this operator's cached tables depend on changes (conv_doe: the PSF); the solvers key
        their denominator / data-spectrum caches on it"""
        if self.__dict__.get("_tables_static"):
            return self.__dict__["_tables_static_value"]
        own = self._own_tables_version()
        out = (own,) + tuple(n.tables_version() for n in self.input_nodes)
        if not self._tables_dynamic():                           # no node below overrides _own_tables_version: the value never changes
            self.__dict__["_tables_static"], self.__dict__["_tables_static_value"] = True, out
        return out

    def _tables_dynamic(self):
        hit = self.__dict__.get("_tables_dyn")
        if hit is None:
            hit = type(self)._own_tables_version is not LinOp._own_tables_version or any(n._tables_dynamic() for n in self.input_nodes)
            self.__dict__["_tables_dyn"] = hit
        return hit

    def _own_tables_version(self):
        return None

    # ---- graph queries ------------------------------------------------------------------------
    @property
    def device(self):
        return self.dummy.device

    # (the graph below a node is fixed once it is built -- input_nodes is assigned in the constructor only --: the walks over it are done
    #  once per node; the solvers ask for them a few dozen times per solve)
    @property
    def variables(self):
        hit = self.__dict__.get("_vars_cache")
        if hit is not None:
            return list(hit)
        found = {}
        for node in self.input_nodes:
            for v in node.variables:
                found[v.uuid] = v
        out = [found[k] for k in sorted(found)]
        self.__dict__["_vars_cache"] = out
        return list(out)

    @property
    def constants(self):
        hit = self.__dict__.get("_consts_cache")
        if hit is not None:
            return list(hit)
        out = []
        for node in self.input_nodes:
            out += node.constants
        self.__dict__["_consts_cache"] = out
        return list(out)

    def is_constant(self):
        return len(self.variables) == 0

    @property
    def value(self):
        """forward evaluation with the current ``Variable.value``s (linop/base.py:109-115)"""
        return self.forward(*[node.value for node in self.input_nodes])

    @property
    def offset(self):
        """the constant part: value with every variable zeroed (linop/base.py:117-129)"""
        saved = [(v, v._value) for v in self.variables]
        try:
            for v, old in saved:
                v._value = torch.zeros_like(old)
            return self.value
        finally:
            for v, old in saved:
                v._value = old

    # ---- derived operators --------------------------------------------------------------------
    @property
    def T(self):
        op = self.clone()
        op.forward, op.adjoint = op.adjoint, op.forward
        return op

    @property
    def gram(self):
        op = self.clone()
        fwd, adj = op.forward, op.adjoint
        op.forward = lambda x: adj(fwd(x))
        op.adjoint = lambda y: fwd(adj(y))
        return op

    def clone(self):
        return _copy.deepcopy(self)

    def unwrap(self, value):
        from .leaf import Placeholder
        if isinstance(value, Placeholder):
            return value.value
        return to_torch_tensor(value, batch=True)

    # ---- operator overloading -----------------------------------------------------------------
    def __add__(self, other):
        from .arith import sum as _sum
        args = []
        for e in (self, cast_to_const(other)):
            args += list(e.input_nodes) if isinstance(e, _sum) else [e]
        return _sum(args)

    def __radd__(self, other):
        return cast_to_const(other) + self

    def __mul__(self, other):
        from .arith import scale
        if not np.isscalar(other):
            raise TypeError("Can only multiply by a scalar constant.")
        return scale(other, self)

    __rmul__ = __mul__

    def __truediv__(self, other):
        from .arith import scale
        if not np.isscalar(other):
            raise TypeError("Can only divide by a scalar constant.")
        return scale(1.0 / other, self)

    __div__ = __truediv__

    def __neg__(self):
        return -1 * self

    def __sub__(self, other):
        return self + (-other)

    def __rsub__(self, other):
        return -self + other

    def __str__(self):
        return self.__class__.__name__

    __array_priority__ = 10000
