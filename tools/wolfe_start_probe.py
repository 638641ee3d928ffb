"""Developer probe (CPU, numpy): Wolfe's min-norm-point on the hull of a query's m nearest members of a bin, started
(a) from the nearest vertex and grown one vertex per major cycle (solve16 up to round 4) or
(b) from the FULL vertex set and pruned (round 5).
Counts insertions / removals / major cycles per problem on the benchmark generator's data, weighted as the 16-lane
kernel pays for them (four problems side by side in a wavefront: the costliest of the four sets the time).
usage: python tools/wolfe_start_probe.py [m] [nprob]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import chbin_amd  # noqa: E402,F401
from chbin_amd import synth  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 15
nprob = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
N, D, B = 20000, 136, 64
X, initial, true = synth.make_synthetic(N, D, B, seed=0)
rng = np.random.default_rng(1)
C_INS, C_REM, C_BETA, C_GRAD = 85, 50, 85, 130   # rough instruction counts of the kernel's pieces


def affine(Q, S, s):
    idx = np.flatnonzero(S)
    A = Q[np.ix_(idx, idx)] + s
    b = np.linalg.solve(A, np.ones(len(idx)))
    beta = np.zeros(len(S))
    beta[idx] = b / b.sum()
    return beta


def wolfe(Q, bulk, order="lane"):
    n = Q.shape[0]
    s = Q.diagonal().max()
    tol = 64 * 2.2e-16 * s
    i0 = int(np.argmin(Q.diagonal()))
    S = np.zeros(n, bool)
    alpha = np.zeros(n)
    alpha[i0] = 1.0
    cost = 0
    st = dict(ins=0, rem=0, major=0)
    if bulk:
        S[:] = True
        cost += n * C_INS
        st["ins"] += n
    else:
        S[i0] = True
        cost += C_INS
        st["ins"] += 1
    for it in range(200):
        # minor cycles
        while True:
            beta = affine(Q, S, s)
            cost += C_BETA
            bad = S & ~(beta > 0)
            if not bad.any():
                alpha = np.where(S, beta, 0.0)
                break
            den = alpha - beta
            ratio = np.where(bad, np.where(den > 0, alpha / np.where(den > 0, den, 1), 0.0), np.inf)
            theta = ratio.min()
            cand = np.flatnonzero(ratio == theta)
            kr = cand[0] if order == "lane" else cand[np.argmin(beta[cand])]
            alpha = np.where(S, alpha + theta * (beta - alpha), 0.0)
            alpha[kr] = 0.0
            S[kr] = False
            cost += C_REM
            st["rem"] += 1
        g = Q @ alpha
        val = alpha @ g
        cost += C_GRAD
        st["major"] += 1
        gm = np.where(S, np.inf, g)
        jb = int(np.argmin(gm))
        if not (gm[jb] < val - tol):
            break
        S[jb] = True
        cost += C_INS
        st["ins"] += 1
    return val, cost, st, int(S.sum())


res = {k: [] for k in ("grow", "bulk", "bulk_neg")}
sup = []
for t in range(nprob):
    j = rng.integers(N)
    c = rng.integers(B) if t % 4 else true[j]
    mem = np.flatnonzero((true == c) & (np.arange(N) != j))
    d = ((X[mem] - X[j]) ** 2).sum(1)
    sel = mem[np.argsort(d)[:m]]
    Y = X[sel] - X[j]
    Q = Y @ Y.T
    v0, c0, s0, k0 = wolfe(Q, False)
    v1, c1, s1, _ = wolfe(Q, True)
    v2, c2, s2, _ = wolfe(Q, True, "neg")
    assert abs(v0 - v1) <= 1e-9 * max(v0, 1e-300) and abs(v0 - v2) <= 1e-9 * max(v0, 1e-300), (v0, v1, v2)
    res["grow"].append((c0, s0["ins"], s0["rem"], s0["major"]))
    res["bulk"].append((c1, s1["ins"], s1["rem"], s1["major"]))
    res["bulk_neg"].append((c2, s2["ins"], s2["rem"], s2["major"]))
    sup.append(k0)
print(f"m = {m}, {nprob} problems, final support mean {np.mean(sup):.2f}")
for k, v in res.items():
    a = np.array(v, float)
    g4 = a[: len(a) // 4 * 4, 0].reshape(-1, 4).max(1).mean()
    print(f"{k:9s}: cost {a[:,0].mean():7.0f} (max of 4: {g4:7.0f})  inserts {a[:,1].mean():5.2f}  removals {a[:,2].mean():5.2f}  major {a[:,3].mean():5.2f}")
