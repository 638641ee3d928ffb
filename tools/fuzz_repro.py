"""Developer probe: replay case <index> of tools/fuzz_fit.py <n> <seed> <mode> and compare the fused path, the
list-based path (CHB_FUSED=0) and the oracle.  usage: python tools/fuzz_repro.py <index> <seed> [mode]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import chbin_amd  # noqa: E402,F401
from chbin_amd import _lib, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

index, seed = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else ""
big, m16 = mode == "big", mode == "m16"
manybins = mode == "manybins"
rng = np.random.default_rng(seed)
for t in range(index + 1):
    N = int(rng.integers(200, 1800))
    D = int(rng.choice([8, 24, 40, 100, 136, 137, 140, 143, 144, 145, 146, 160, 161, 200]))
    B = int(rng.integers(1, 24))
    m = int(rng.choice([1, 2, 3, 5, 5, 5, 8, 9, 15, 16]))
    S = 1 if D < 140 else (5 if D < 146 else 10)
    iters = int(rng.integers(1, 6))
    batch = int(rng.choice([0, 1, 7, 64, 100, 257, 1000, 4096]))
    sigma = float(rng.choice([1.5e-3, 4e-3, 9e-3]))
    mix = float(rng.choice([0.0, 0.3, 0.6, 0.9]))
    n_seed = int(rng.integers(1, 12))
    if big:
        N = int(rng.integers(3000, 7000)); B = int(rng.integers(2, 7)); m = int(rng.choice([3, 5, 5, 8]))
        iters = int(rng.integers(1, 4)); batch = int(rng.choice([0, 512, 2048])); n_seed = int(rng.integers(5, 40))
        D = int(rng.choice([100, 136, 140, 146]))
        S = 1 if D < 140 else (5 if D < 146 else 10)
    if m16:
        N = int(rng.integers(200, 1100)); m = int(rng.integers(6, 17)); D = int(rng.choice([24, 40, 100, 136, 140, 146, 160]))
        S = 1 if D < 140 else (5 if D < 146 else 10)
        iters = int(rng.integers(1, 4)); n_seed = int(rng.integers(1, 24))
    if manybins:
        N = int(rng.integers(2000, 5000)); B = int(rng.integers(65, 400)); m = int(rng.choice([1, 3, 5, 5]))
        D = int(rng.choice([100, 136, 140, 146])); S = 1 if D < 140 else (5 if D < 146 else 10)
        iters = int(rng.integers(1, 3)); batch = int(rng.choice([0, 300, 1000])); n_seed = int(rng.integers(1, 4))
    metric = str(rng.choice(["convex", "convex", "convex", "affine"]))
    if m > D or D < 24 or (D < 40 and m > 8):   # (round 4, case `120 91 m16` #101: 14 vertices in D = 24 straddle orth's cutoff too)
        metric = "convex"
    X, initial, _ = synth.make_synthetic(N, D, B, S=min(S, max(D - 4, 1)), seed=int(rng.integers(1 << 30)), sigma=sigma,
                                         mix=mix, n_seed=n_seed)
    if rng.random() < 0.2 and m <= 8:
        X = X * float(10.0 ** rng.integers(-6, 7))
    perms = synth.draw_permutations(initial, iters, seed=int(rng.integers(1 << 30)))
print(f"case {index}: N={N} D={D} B={B} m={m} iters={iters} batch={batch} sigma={sigma} mix={mix} seeds={n_seed} {metric}")
want, its_o, ch_o = O.fit_cluster(X, B, initial, perms, m, iters, metric=metric)
res = {}
for name, env in (("fused", {}), ("lists", {"CHB_FUSED": "0"}), ("brute", {"CHB_PREFILTER": "0"})):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    c = _lib.Context(0)
    for k, v in old.items():
        if v is None:
            del os.environ[k]
        else:
            os.environ[k] = v
    c.set_metric(metric)
    c.set_samples(X)
    got, its, ch, mind = c.fit_cluster(B, initial, perms, m, iters, batch=batch, want_min_dist=True)
    res[name] = (got, mind)
    print(f"{name}: sweeps {its}/{its_o}, labels differing from the oracle {int((got != want).sum())}, changes {list(ch)} vs {list(ch_o)}")
    c.close()
# the first sweep against the oracle's: where do the fused path's winning distances leave the oracle's?
for name in res:
    got1 = None
c = _lib.Context(0)
c.set_metric(metric)
c.set_samples(X)
g1, _, _, md1 = c.fit_cluster(B, initial, perms[:1], m, 1, batch=batch, want_min_dist=True)
c.close()
lab_o, md_o = O.sweep(X, B, initial, perms[0], m, metric=metric)
dif = np.flatnonzero(g1[perms[0]] != lab_o[perms[0]])
print("sweep 1: labels differing", len(dif), "first positions", dif[:5])
err = np.abs(md1[perms[0]] - md_o)
print("sweep 1: max |winning distance - oracle| over positions before the first differing one:",
      float(err[: (dif[0] if len(dif) else len(err))].max()) if len(err) else 0.0)
if len(dif):
    i = int(dif[0])
    print("at the first differing position: GPU distance", md1[perms[0][i]], "oracle", md_o[i], "labels", g1[perms[0][i]], lab_o[perms[0][i]])
# the hull problems of the first position whose winning distance leaves the oracle's, bin by bin
import scipy.linalg  # noqa: E402
from chbin_amd import clustering  # noqa: E402
bad = np.flatnonzero(err > 1e-7)
if len(bad):
    i = int(bad[0])
    lab = initial.copy()
    if i > 0:
        lab, _ = O.sweep(X, B, lab, perms[0][:i], m, metric=metric)
    j = int(perms[0][i])
    print(f"position {i} (sample {j}): GPU {md1[j]} oracle {md_o[i]}")
    for c in range(B):
        members = np.flatnonzero(lab == c)
        members = members[members != j]
        if len(members) == 0:
            continue
        d = np.sqrt(((X[members] - X[j]) ** 2).sum(axis=1))
        order = np.lexsort((members, d))[:m]
        P = X[members[order]]
        mean = P.mean(axis=0)
        basis = scipy.linalg.orth((P - mean).T)
        proj = basis @ np.linalg.inv(basis.T @ basis) @ basis.T
        ref = np.linalg.norm((np.eye(proj.shape[0]) - proj) @ (X[j] - mean))
        sv = np.linalg.svd(P - mean, compute_uv=False)
        print(f"  bin {c}: {len(order)} vertices, numpy formula {ref:.12g}, oracle {O.affine_hull_distance(X[j], P):.12g}, "
              f"GPU {clustering.calculate_distance(X[j], P, 'quadprog', metric):.12g}, convex (GPU) "
              f"{clustering.calculate_distance(X[j], P, 'quadprog', 'convex'):.12g}; singular values {sv[0]:.3g} .. {sv[-2]:.3g}, {sv[-1]:.3g}")
