"""VERDICT r4 item 2, sized before building (CPU / numpy): if the threshold sweep of the shortlist kernel is replaced by a
POOL -- for every ordered pair (home bin h, bin c) the P members of c nearest to the centre of h, and tau(j, c) = the m-th
smallest distance from query j to the pool of its home bin h = nearest centre -- how many members of c pass d <= tau?
(Any m members of c give a valid upper bound of the m-th nearest distance; today's two-sweep build admits ~5.3 at m = 5.)

usage: python tools/pool_tau_probe.py N D B [m] [mix] [sigma] [frac labelled] [queries]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chbin_amd  # noqa: E402,F401
from chbin_amd import synth  # noqa: E402

av = sys.argv[1:]
N, D, B = int(av[0]), int(av[1]), int(av[2])
m = int(av[3]) if len(av) > 3 else 5
mix = float(av[4]) if len(av) > 4 else 0.0
sigma = float(av[5]) if len(av) > 5 else 1.5e-3
frac = float(av[6]) if len(av) > 6 else 1.0
NQ = int(av[7]) if len(av) > 7 else 192
S = 1 if D <= 136 else (5 if D == 140 else 10)
X, initial, true = synth.make_synthetic(N, D, B, S=S, seed=0, mix=mix, sigma=sigma)
rng = np.random.default_rng(1)
centers = np.stack([X[initial == c].mean(axis=0) for c in range(B)])          # as the library: mean of the seeds
# the labelled set in the middle of sweep 1: the seeds + a fraction of the rest under their true label
lab = initial.copy()
rest = np.flatnonzero(initial < 0)
take = rest[rng.random(len(rest)) < frac]
lab[take] = true[take]
members = [np.flatnonzero(lab == c) for c in range(B)]
# squared distance of every sample to every centre (the library's per-fit qn table)
x2 = (X * X).sum(1)
qn = x2[:, None] - 2.0 * X @ centers.T + (centers * centers).sum(1)[None, :]
queries = rng.choice(np.flatnonzero(lab < 0) if frac < 1.0 else rest, NQ, replace=False)
home = qn[queries].argmin(1)
second = np.argsort(qn[queries], axis=1)[:, 1]
print(f"N={N} D={D} B={B} m={m} mix={mix} sigma={sigma} labelled fraction {frac}: members per bin ~{int(np.mean([len(v) for v in members]))}, "
      f"queries {NQ}, home == true bin for {np.mean(home == true[queries]):.3f}")
for P in (16, 32, 64):
    for keyname in ("nearest to centre h", "projection on mu_h - mu_c"):
        cnt_ok, cnt_wrong, cnt_home = [], [], []
        for qi, j in enumerate(queries):
            for c in rng.choice(B, min(B, 12), replace=False).tolist() + [int(home[qi])]:
                mem = members[c]
                mem = mem[mem != j]
                if len(mem) <= m:
                    continue
                d = np.linalg.norm(X[mem] - X[j], axis=1)

                def pool_tau(h):
                    if keyname.startswith("nearest") or h == c:
                        key = qn[mem, h]
                    else:
                        key = qn[mem, h] - qn[mem, c]
                    pool = np.argpartition(key, min(P, len(mem)) - 1)[:P]
                    return np.sort(d[pool])[m - 1]
                n_ok = int((d <= pool_tau(int(home[qi]))).sum())
                (cnt_home if c == home[qi] else cnt_ok).append(n_ok)
                if c != home[qi] and c != second[qi]:
                    cnt_wrong.append(int((d <= pool_tau(int(second[qi]))).sum()))     # the pool of another home bin

        def rep(v):
            v = np.asarray(v)
            return f"mean {v.mean():7.2f}  p99 {np.percentile(v, 99):6.0f}  >128: {np.mean(v > 128):.4f}"
        print(f"  P={P:3d} key = {keyname:28s} other bins: {rep(cnt_ok)} | own bin: {rep(cnt_home)} | pool of the 2nd-nearest home: {rep(cnt_wrong)}")
