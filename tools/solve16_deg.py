"""Developer probe (GPU): ONE of the near-degenerate trials of tools/solve16_cases.py (its number), e.g. under a tracing build."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import chbin_amd  # noqa: E402,F401
from chbin_amd import _lib  # noqa: E402
from oracle import oracle as O  # noqa: E402

T = int(sys.argv[1])
rngd = np.random.default_rng(7)
for trial in range(T + 1):
    D = int(rngd.integers(2, 6)); m = int(rngd.integers(6, 17))
    P = rngd.standard_normal((m, D))
    k = int(rngd.integers(2, D + 1))
    eps_off = 10.0 ** rngd.uniform(-7, -1)
    P[k:] = rngd.dirichlet(np.ones(k), size=m - k) @ P[:k] + eps_off * rngd.standard_normal((m - k, D))
    x = P.mean(0) + 0.5 * rngd.standard_normal(D)
ctx = _lib.Context(0)
d, alpha = ctx.hull_distance_points(x, P, want_alpha=True)
print("gpu", d, "oracle", O.convex_hull_distance(np.zeros(D), P - x), "D", D, "m", m)
print("alpha", alpha.tolist())
