"""Leaves of the expression graph: ``Variable``, ``Constant``, ``Placeholder``
(reference dprox/linop/variable.py:8-100, constant.py:7-96, placeholder.py:4-22)."""
import itertools

import torch

from .node import LinOp

_var_ids = itertools.count()


def device_f32(t, device):
    """values enter the HIP kernels as fp32 (complex64 for complex data) on the graph's device"""
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t)
    dt = torch.complex64 if t.is_complex() else torch.float32
    return t.detach().to(device=device, dtype=dt) if not t.requires_grad else t.to(device=device, dtype=dt)


class Variable(LinOp):
    def __init__(self, shape=None, value=None, name=None):
        super().__init__([])
        self.uuid = next(_var_ids)       # creation order, like the reference's time-based uuid1
        self._value = value
        self.shape = shape
        self.varname = name
        self.initval = None

    def forward(self, inputs, **kwargs):
        return inputs

    def adjoint(self, inputs, **kwargs):
        return inputs

    def is_diag(self, freq=False):
        return True

    def get_diag(self, ref, freq=False):
        return torch.ones(ref.shape)

    @property
    def variables(self):
        return [self]

    @property
    def value(self):
        return self._value.to(self.device)

    @value.setter
    def value(self, val):
        self._value = val

    def __setattr__(self, name, val):
        # (`_value` / `value` are plain attributes holding a tensor or None: nn.Module.__setattr__ walks its parameter / buffer / module
        #  tables for every assignment -- the solvers assign the iterate to the variable twice per call)
        if name == "_value" or name == "value":
            object.__setattr__(self, "_value", val)
        else:
            super().__setattr__(name, val)

    def norm_bound(self, input_mags):
        return 1.0

    def __repr__(self):
        return f"Variable(id={self.uuid}, shape={self.shape}, value={'None' if self._value is None else 'somevalue'})"


class Constant(LinOp):
    def __init__(self, value):
        super().__init__([])
        if value is not None and not isinstance(value, torch.Tensor):
            value = torch.tensor(value)
        self._value = value
        self._version = 0
        self._dev_cache = None

    def forward(self, *value, **kwargs):
        return self.value

    def adjoint(self, value, **kwargs):
        return None          # contributes nothing to any variable (reference returns value*0)

    def is_diag(self, freq=False):
        return True

    def get_diag(self, ref=None, freq=False):
        return {}

    @property
    def variables(self):
        return []

    @property
    def constants(self):
        return [self]

    @property
    def value(self):
        key = (self._version, str(self.device))
        if self._dev_cache is None or self._dev_cache[0] != key:
            self._dev_cache = (key, device_f32(self._value, self.device))
        return self._dev_cache[1]

    def norm_bound(self, input_mags):
        return 0.0

    def __repr__(self):
        return "Constant(value=%s)" % ("somevalue" if self._value is not None else "None")


class Placeholder(Constant):
    """a constant whose value is assigned later; watchers fire on assignment (placeholder.py:13-22)"""

    def __init__(self, default=None):
        super().__init__(default)
        self.watchers = []

    @property
    def value(self):
        return Constant.value.fget(self)

    @value.setter
    def value(self, val):
        self._value = val
        self._version += 1
        for w in self.watchers:
            w(val)

    def __setattr__(self, name, val):
        # nn.Module.__setattr__ would swallow `ph.value = nn.Parameter(...)`;
        # route every assignment of `value` through the property so watchers always fire.
        if name == "value":
            type(self).value.fset(self, val)
        else:
            super().__setattr__(name, val)

    def change(self, fn):
        self.watchers.append(fn)
