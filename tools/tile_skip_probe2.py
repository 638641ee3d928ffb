"""How much of the shortlist stage's tile stream could EXACT tile skipping remove, as a kernel would see it?  numpy only.
Members of a bin bucketed into 16 shells by their distance from the bin's centre (what a CSR key (bin, shell) gives
for free; unordered inside a shell), periphery first, 32-row tiles with (centre, radius); a tile is skippable for a
query when d(query, centre) - radius exceeds the threshold: the exact m-th nearest distance (sweep 1), or the running
m-th best of a tile-best pass in stream order (sweep 0).  A wavefront skips a tile only if all its 32 queries do:
random queries of a batch, or queries sorted by their nearest bin centre.
usage: python tools/tile_skip_probe2.py N D B m     (D = 136: one coverage column, 140: five, 146: ten)"""
import os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
import chbin_amd
from chbin_amd import synth
N, D, B, m = (int(x) for x in sys.argv[1:5])
S = 1 if D <= 136 else (5 if D == 140 else 10)
X, initial, true = synth.make_synthetic(N, D, B, S=S, seed=0)
rng = np.random.default_rng(0)
centers = np.stack([X[true == c].mean(axis=0) for c in range(B)])
NSH = 16
tot = {"sweep1_query": [0,0], "sweep0_query": [0,0], "sweep1_wave_random": [0,0], "sweep1_wave_sorted": [0,0], "sweep0_wave_sorted":[0,0]}
near = np.argmin(((X[:, None, :] - centers[None, :, :])**2).sum(-1) if N*B < 3e7 else np.stack([((X-centers[c])**2).sum(1) for c in range(B)],1), axis=1)
for c in rng.choice(B, 5, replace=False):
    memb = X[true == c]
    z = memb - centers[c]
    nr = np.linalg.norm(z, axis=1)
    shell = np.minimum((NSH * nr / nr.max()).astype(int), NSH - 1)
    order = np.argsort(-shell, kind="stable")          # periphery first, unordered inside a shell
    rows = memb[order]
    T = (len(rows) + 31) // 32
    ctr = np.stack([rows[t*32:(t+1)*32].mean(0) for t in range(T)])
    rad = np.array([np.linalg.norm(rows[t*32:(t+1)*32] - ctr[t], axis=1).max() for t in range(T)])
    def per_query(q):
        d = np.linalg.norm(rows - X[q], axis=1)
        if true[q] == c: d[d == 0] = np.inf
        tau = np.sort(d)[m - 1]
        lb = np.linalg.norm(ctr - X[q], axis=1) - rad
        s1 = lb > tau
        # sweep 0 with running tau: tile-best insertion in stream order
        best = []
        s0 = np.zeros(T, bool)
        for t in range(T):
            cur = np.sort(best)[m-1] if len(best) >= m else np.inf
            if lb[t] > cur: s0[t] = True; continue
            dd = d[t*32:(t+1)*32]
            best.append(dd.min()); best = sorted(best)[:m] if len(best) > m else best
        return s1, s0
    qs = rng.choice(N, 64, replace=False)
    S1 = []; 
    for q in qs:
        s1, s0 = per_query(q); S1.append(s1)
        tot["sweep1_query"][0] += s1.sum(); tot["sweep1_query"][1] += T
        tot["sweep0_query"][0] += s0.sum(); tot["sweep0_query"][1] += T
    S1 = np.array(S1)
    for w in range(2):
        tot["sweep1_wave_random"][0] += S1[w*32:(w+1)*32].all(0).sum(); tot["sweep1_wave_random"][1] += T
    # waves of 32 queries that share their nearest centre (other than c)
    for cc in rng.choice([x for x in range(B) if x != c], 3, replace=False):
        qq = np.flatnonzero(near == cc)[:32]
        r = [per_query(q) for q in qq]
        tot["sweep1_wave_sorted"][0] += np.array([a for a, b in r]).all(0).sum(); tot["sweep1_wave_sorted"][1] += T
        tot["sweep0_wave_sorted"][0] += np.array([b for a, b in r]).all(0).sum(); tot["sweep0_wave_sorted"][1] += T
print(f"N={N} D={D} B={B}: {NSH} shells, tiles per bin ~{N//B//32}")
for k,(a,b) in tot.items(): print(f"  {k:22s} {100.0*a/b:6.2f} % of tiles skippable")
