"""Developer probe (GPU): the hull-distance cases of tests/test_gpu_parity.py::test_hull_distance_16_lane_solver_both_starts,
every case against the oracle (shifted to the query), with the weights the GPU returned for the ones that differ.
usage: python tools/solve16_cases.py [D ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import chbin_amd  # noqa: E402,F401
from chbin_amd import _lib  # noqa: E402
from oracle import oracle as O  # noqa: E402


def cases(D):
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
        yield t, m, kind, x, P


ctx = _lib.Context(0)
for D in [int(a) for a in sys.argv[1:]] or [3, 7, 40, 136]:
    bad = 0
    for t, m, kind, x, P in cases(D):
        d, alpha = ctx.hull_distance_points(x, P, want_alpha=True)
        d_or = O.convex_hull_distance(np.zeros(D), P - x)
        scale = np.linalg.norm(P - x, axis=1).max()
        if abs(d - d_or) > 1e-9 * max(scale, 1.0) + 1e-7 * scale * (d_or < 1e-6 * scale):
            bad += 1
            Y = P - x
            Q = Y @ Y.T
            g = Q @ alpha
            print(f"D={D} t={t} m={m} kind={kind}: gpu {d:.12f} oracle {d_or:.12f}  support {np.flatnonzero(alpha > 0).tolist()}")
            print("     alpha", np.round(alpha, 6).tolist())
            print("     gradient - val (negative: an improving vertex was left out):", np.round(g - alpha @ g, 9).tolist())
    print(f"D={D}: {bad} of 132 differ")

# ---- near-degenerate vertex sets (nearly coplanar / collinear: supports with cond 1e2 .. 1e13), by thickness decade
rngd = np.random.default_rng(7)
dec = {}
for trial in range(3000):
    D = int(rngd.integers(2, 6)); m = int(rngd.integers(6, 17))
    P = rngd.standard_normal((m, D))
    k = int(rngd.integers(2, D + 1))
    eps_off = 10.0 ** rngd.uniform(-8, -1)
    P[k:] = rngd.dirichlet(np.ones(k), size=m - k) @ P[:k] + eps_off * rngd.standard_normal((m - k, D))
    x = P.mean(0) + 0.5 * rngd.standard_normal(D)
    d = ctx.hull_distance_points(x, P)
    d = d[0] if isinstance(d, tuple) else d
    # yardstick: the enumerator where it is affordable (exact to rounding), else Goldfarb-Idnani on the shifted problem
    truth = O.enum_hull_distance(np.zeros(D), P - x) if m <= 12 else O.convex_hull_distance(np.zeros(D), P - x)
    scale = np.linalg.norm(P - x, axis=1).max()
    err = abs(d - truth) / max(scale, 1e-300)
    key = (int(np.floor(np.log10(eps_off))), "m<=8" if m <= 8 else "m>8", "enum" if m <= 12 else "gi")
    w = dec.setdefault(key, [0, 0.0])
    w[0] += 1; w[1] = max(w[1], err)
for key in sorted(dec):
    print(f"thickness 1e{key[0]:+d} {key[1]:5s} vs {key[2]:4s}: {dec[key][0]:4d} cases, worst error {dec[key][1]:.2e} of the scale")
