"""Regenerates tests/golden/affine_16_vertices.npz: the hull problem on which a Gram-Schmidt restatement of
hull_distance.py:69-87 (cutoff on residual norms) left scipy.linalg.orth's value -- case 11 of
`tools/fuzz_fit.py 60 11 m16` (N=329, D=40, B=2, m=16, affine metric), position 12 of sweep 1, bin 1.
Inputs: the query x and its 16 nearest members P of that bin; expected: the reference formula evaluated
literally with numpy / scipy (the reference's own dependencies).  CPU only."""
import os
import sys

import numpy as np
import scipy.linalg

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import chbin_amd  # noqa: E402,F401
from chbin_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

rng = np.random.default_rng(11)
for t in range(12):   # replay tools/fuzz_fit.py's draws up to case 11 (mode m16)
    N = int(rng.integers(200, 1800)); D = int(rng.choice([8, 24, 40, 100, 136, 137, 140, 143, 144, 145, 146, 160, 161, 200]))
    B = int(rng.integers(1, 24)); m = int(rng.choice([1, 2, 3, 5, 5, 5, 8, 9, 15, 16]))
    S = 1 if D < 140 else (5 if D < 146 else 10)
    iters = int(rng.integers(1, 6)); batch = int(rng.choice([0, 1, 7, 64, 100, 257, 1000, 4096]))
    sigma = float(rng.choice([1.5e-3, 4e-3, 9e-3])); mix = float(rng.choice([0.0, 0.3, 0.6, 0.9]))
    n_seed = int(rng.integers(1, 12))
    N = int(rng.integers(200, 1100)); m = int(rng.integers(6, 17)); D = int(rng.choice([24, 40, 100, 136, 140, 146, 160]))
    S = 1 if D < 140 else (5 if D < 146 else 10)
    iters = int(rng.integers(1, 4)); n_seed = int(rng.integers(1, 24))
    metric = str(rng.choice(["convex", "convex", "convex", "affine"]))
    if m > D or D < 24:
        metric = "convex"
    X, initial, _ = synth.make_synthetic(N, D, B, S=min(S, max(D - 4, 1)), seed=int(rng.integers(1 << 30)), sigma=sigma,
                                         mix=mix, n_seed=n_seed)
    if rng.random() < 0.2 and m <= 8:
        X = X * float(10.0 ** rng.integers(-6, 7))
    perms = synth.draw_permutations(initial, iters, seed=int(rng.integers(1 << 30)))
assert (N, D, B, m, metric) == (329, 40, 2, 16, "affine"), (N, D, B, m, metric)
lab, _ = O.sweep(X, B, initial, perms[0][:12], m, metric=metric)
j = int(perms[0][12])
members = np.flatnonzero(lab == 1)
members = members[members != j]
d = np.sqrt(((X[members] - X[j]) ** 2).sum(axis=1))
P = X[members[np.lexsort((members, d))[:m]]]
x = X[j]
mean = P.mean(axis=0)
basis = scipy.linalg.orth((P - mean).T)
proj = basis @ np.linalg.inv(basis.T @ basis) @ basis.T
want = np.linalg.norm((np.eye(proj.shape[0]) - proj) @ (x - mean))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "affine_16_vertices.npz")
np.savez(out, x=x, P=P, expected=np.float64(want), rank=np.int64(basis.shape[1]))
print("wrote", out, "expected", want, "rank", basis.shape[1], "oracle", O.affine_hull_distance(x, P))
