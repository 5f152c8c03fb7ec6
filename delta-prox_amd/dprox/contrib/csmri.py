"""The solver class of the reference's CS-MRI pipelines (reference dprox/contrib/csmri.py:156-171): ADMM in the
(x = prior output, z = data-term output, u) ordering on a complex iterate.  The datasets / RL environment of that
file are outside the hot path."""
from .. import _ops as ops
from ..algo.splitting import ADMM


class CustomADMM(ADMM):
    def _iter(self, state, rho, lam):
        x, z, u = state
        x = [x]
        z = z[0]
        n = len(self.psi_fns)
        for i, fn in enumerate(self.psi_fns):
            x[i] = fn.prox(ops.clincomb([(1.0, z), (-1.0, u[i])], out_complex=z.is_complex() or u[i].is_complex()), lam=lam[fn])
        b = [ops.clincomb([(1.0, x[i]), (1.0, u[i])], out_complex=x[i].is_complex() or u[i].is_complex()) for i in range(n)]
        z = self.least_square.solve(b, rho)
        for i, fn in enumerate(self.psi_fns):
            u[i] = ops.clincomb([(1.0, u[i]), (1.0, x[i]), (-1.0, z)], out_complex=True)
        return x[0], [z], u
