"""Developer probe (GPU): ONE case of tools/solve16_cases.py (D, t), e.g. under a tracing build of the solver (CHBIN_LIB)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

D, T = int(sys.argv[1]), int(sys.argv[2])
import chbin_amd  # noqa: E402,F401
from chbin_amd import _lib  # noqa: E402
from oracle import oracle as O  # noqa: E402

rng = np.random.default_rng(500 + D)
for t in range(132):
    m = 6 + t % 11
    kind = (t // 11) % 6
    P = rng.standard_normal((m, D))
    if kind == 0:
        x = 0.3 * rng.standard_normal(D)
    elif kind == 1:
        x = 40.0 * np.ones(D) + rng.standard_normal(D)
    elif kind == 2:
        x = rng.dirichlet(np.ones(m)) @ P
    elif kind == 3:
        P[m - 1] = P[0]; P[m - 2] = P[1]
        x = 0.3 * rng.standard_normal(D)
    elif kind == 4:
        P[3:] = rng.dirichlet(np.ones(3), size=m - 3) @ P[:3] + 1e-3 * rng.standard_normal((m - 3, D))
        x = P.mean(0) + 0.5 * rng.standard_normal(D)
    else:
        P *= 1e-6; x = 1e-6 * 0.3 * rng.standard_normal(D) + 5.0
        P += 5.0
    if t == T:
        break
ctx = _lib.Context(0)
d, alpha = ctx.hull_distance_points(x, P, want_alpha=True)
print("gpu", d, "oracle", O.convex_hull_distance(np.zeros(D), P - x))
print("alpha", alpha.tolist())
