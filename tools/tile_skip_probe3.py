"""Round 4 probe (numpy only): would a per-tile BOX in the coverage columns' subspace skip more member tiles than the
norm-shell bound (built) or the (centre, radius) bound of tile_skip_probe2.py?
Members of a bin: norm shells (16 / 32), inside a shell ordered by a key of their coverage columns (first principal
direction of the bin's coverage block, or a Morton code of the two leading ones); 32-row tiles; per tile the box
[min, max] of every coverage column and the (centre, radius) ball in all columns.  Lower bounds of d(query, member of tile):
  norm  : | ||z_q|| - ||z_p|| | with the tile's norm range                       (what round 3 built, per tile)
  ball  : d(q, centre) - radius
  box   : distance of the query's coverage columns to the tile's box
A tile is skippable when the bound exceeds the query's exact m-th nearest distance in the bin (sweep 1).  Per query, and per
wavefront of 32 queries that share their nearest bin centre.
usage: python tools/tile_skip_probe3.py N D B m"""
import sys
import numpy as np
sys.path.insert(0, '/root/repo')
import chbin_amd
from chbin_amd import synth
N, D, B, m = (int(x) for x in sys.argv[1:5])
S = 1 if D <= 136 else (5 if D == 140 else 10)
X, initial, true = synth.make_synthetic(N, D, B, S=S, seed=0)
rng = np.random.default_rng(0)
centers = np.stack([X[true == c].mean(axis=0) for c in range(B)])
near = np.stack([((X - centers[c]) ** 2).sum(1) for c in range(B)], 1).argmin(1)
cov = slice(D - S, D)
res = {}
def add(k, a, b):
    r = res.setdefault(k, [0, 0]); r[0] += a; r[1] += b
for c in rng.choice(B, 4, replace=False):
    memb = X[true == c]
    z = memb - centers[c]
    nr = np.linalg.norm(z, axis=1)
    for NSH, order_kind in ((32, "none"), (32, "pc1"), (8, "pc1"), (4, "morton"), (1, "morton")):
        shell = np.minimum((NSH * nr / nr.max()).astype(int), NSH - 1)
        zc = z[:, cov]
        if order_kind == "none":
            sub = np.zeros(len(z))
        elif order_kind == "pc1":
            u, s_, vt = np.linalg.svd(zc - zc.mean(0), full_matrices=False)
            sub = zc @ vt[0]
        else:
            u, s_, vt = np.linalg.svd(zc - zc.mean(0), full_matrices=False)
            p2 = (zc @ vt[:2].T)
            q2 = np.stack([np.searchsorted(np.sort(p2[:, i]), p2[:, i]) * 1024 // len(p2) for i in range(2)], 1)
            sub = np.zeros(len(z), dtype=np.int64)
            for bit in range(10):
                sub |= ((q2[:, 0] >> bit) & 1) << (2 * bit) | ((q2[:, 1] >> bit) & 1) << (2 * bit + 1)
        order = np.lexsort((sub, -shell))
        rows = memb[order]; zr = rows - centers[c]
        T = (len(rows) + 31) // 32
        sl = [slice(t * 32, (t + 1) * 32) for t in range(T)]
        nmin = np.array([np.linalg.norm(zr[s], axis=1).min() for s in sl]); nmax = np.array([np.linalg.norm(zr[s], axis=1).max() for s in sl])
        ctr = np.stack([rows[s].mean(0) for s in sl]); rad = np.array([np.linalg.norm(rows[s] - ctr[t], axis=1).max() for t, s in enumerate(sl)])
        bmin = np.stack([rows[s][:, cov].min(0) for s in sl]); bmax = np.stack([rows[s][:, cov].max(0) for s in sl])
        def bounds(q):
            d = np.linalg.norm(rows - X[q], axis=1)
            if true[q] == c: d[d == 0] = np.inf
            tau = np.sort(d)[m - 1]
            nq = np.linalg.norm(X[q] - centers[c])
            lb_n = np.maximum(nq - nmax, nmin - nq)
            lb_b = np.linalg.norm(ctr - X[q], axis=1) - rad
            xq = X[q, cov]
            lb_x = np.linalg.norm(np.maximum(0, np.maximum(bmin - xq, xq - bmax)), axis=1)
            return lb_n > tau, lb_b > tau, lb_x > tau
        tag = f"{NSH:2d} shells / {order_kind:6s}"
        for cc in rng.choice([x for x in range(B) if x != c], 3, replace=False):
            qq = np.flatnonzero(near == cc)[:32]
            r = [bounds(q) for q in qq]
            for i, nm in enumerate(("norm", "ball", "box")):
                a = np.array([x[i] for x in r])
                add((tag, nm, "query"), a.sum(), a.size); add((tag, nm, "wave"), a.all(0).sum(), T)
            a = np.array([x[0] | x[1] | x[2] for x in r]); add((tag, "any", "wave"), a.all(0).sum(), T)
            a = np.array([x[0] | x[2] for x in r]); add((tag, "norm|box", "wave"), a.all(0).sum(), T)
        # queries of the bin itself (the home tiles: the longest work items)
        qq = np.flatnonzero(near == c)[:32]
        r = [bounds(q) for q in qq]
        a = np.array([x[0] | x[2] for x in r]); add((tag, "norm|box", "home wave"), a.all(0).sum(), T)
        a = np.array([x[0] for x in r]); add((tag, "norm", "home wave"), a.all(0).sum(), T)
print(f"N={N} D={D} B={B} m={m}: tiles per bin ~{N // B // 32}")
for k in sorted(res): print(f"  {k[0]}  {k[1]:9s} {k[2]:10s} {100.0 * res[k][0] / res[k][1]:6.2f} % of tiles skippable")
