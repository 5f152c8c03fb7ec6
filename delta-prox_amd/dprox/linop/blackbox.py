"""User-supplied operators: ``LinOpFactory(forward, adjoint, diag, norm_bound)``
(reference dprox/linop/blackbox.py:4-72).  The callables receive ``(input, step=...)``."""
from .node import LinOp


class BlackBox(LinOp):
    def __init__(self, *args, forward=None, adjoint=None, diag=None, norm_bound=None):
        self._forward, self._adjoint, self._diag, self._norm_bound = forward, adjoint, diag, norm_bound
        super().__init__(args)

    def forward(self, *inputs, **kwargs):
        return self._forward(*inputs, step=self.step)

    def adjoint(self, *inputs, **kwargs):
        return self._adjoint(*inputs, step=self.step)

    def norm_bound(self, input_mags):
        if self._norm_bound is None:
            return NotImplemented
        return self._norm_bound * input_mags[0]

    def is_gram_diag(self, freq=False):
        return self._diag is not None

    def get_diag(self, x, freq=False):
        return self._diag(x, self.step)


def LinOpFactory(forward, adjoint, diag=None, norm_bound=None):
    def make(*args):
        return BlackBox(*args, forward=forward, adjoint=adjoint, diag=diag, norm_bound=norm_bound)
    return make
