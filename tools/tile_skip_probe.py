"""Would exact tile skipping remove work on this data?  CPU / numpy only.

For a sample of (query, bin) pairs of the benchmark generator: the members of the bin are ordered by a spatial key, cut
into 32-row tiles with (centre, radius), and a tile could be skipped when  d(query, centre) - radius > tau,  tau = the
exact m-th nearest member distance (the best threshold any first pass could deliver).  Three orders are tried: by the
norm of the bin-centred member, by its projection on the bin's first principal direction, and -- an oracle no pack can
have, since it differs per pair of bins -- by the projection on the direction from the bin's centre to the query's bin.

usage: python tools/tile_skip_probe.py [N] [D] [B] [m]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chbin_amd  # noqa: E402,F401
from chbin_amd import synth  # noqa: E402

N, D, B, m = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (100_000, 136, 64, 5)))
S = 1 if D <= 136 else (5 if D == 140 else 10)
X, initial, true = synth.make_synthetic(N, D, B, S=S, seed=0)
rng = np.random.default_rng(0)
centers = np.stack([X[true == c].mean(axis=0) for c in range(B)])


def tiles_of(members, key):
    order = np.argsort(key)
    rows = members[order]
    out = []
    for t in range(0, len(rows), 32):
        blk = rows[t:t + 32]
        ctr = blk.mean(axis=0)
        out.append((ctr, np.linalg.norm(blk - ctr, axis=1).max()))
    return out


res = {"norm": [0, 0], "pc1": [0, 0], "towards query's bin (oracle order)": [0, 0]}
for c in rng.choice(B, 6, replace=False):
    memb = X[true == c]
    z = memb - centers[c]
    u, s_, vt = np.linalg.svd(z[rng.choice(len(z), min(len(z), 600), replace=False)], full_matrices=False)
    keyed = {"norm": tiles_of(memb, np.linalg.norm(z, axis=1)), "pc1": tiles_of(memb, z @ vt[0])}
    for q in rng.choice(N, 40, replace=False):
        d = np.linalg.norm(memb - X[q], axis=1)
        if true[q] == c:
            d = np.sort(d)[1:]          # (the query itself is not a member)
        tau = np.sort(d)[m - 1]
        dirq = centers[true[q]] - centers[c]
        keyed["towards query's bin (oracle order)"] = tiles_of(memb, z @ dirq) if true[q] != c else keyed["pc1"]
        for name, tl in keyed.items():
            skip = sum(1 for ctr, rad in tl if np.linalg.norm(X[q] - ctr) - rad > tau)
            res[name][0] += skip
            res[name][1] += len(tl)
print(f"N={N} D={D} B={B} m={m}: members per bin ~{N // B}, tiles per bin ~{N // B // 32}")
for name, (skip, tot) in res.items():
    print(f"  order by {name:38s}: {skip} of {tot} tiles skippable = {100.0 * skip / tot:.2f} %")
