"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the Delta-Prox ADMM/PGD hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and there only as the checker / the timed CPU baseline -- never as the thing shipped.
The product (``delta-prox_amd/``) must not import this package.
"""
from .dprox_oracle import *  # noqa: F401,F403
