"""Developer probe: the same fit with and without the threshold pools (two contexts): labels must be identical.
usage: python tools/pool_ab.py N D B [m] [sweeps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chbin_amd  # noqa: E402,F401
from chbin_amd import _lib, synth  # noqa: E402

N, D, B = (int(v) for v in sys.argv[1:4])
m = int(sys.argv[4]) if len(sys.argv) > 4 else 5
sweeps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
S = 1 if D <= 136 else (5 if D == 140 else 10)
X, initial, true = synth.make_synthetic(N, D, B, S=S, seed=0)
perms = synth.draw_permutations(initial, sweeps, seed=0)
out = {}
for pool in ("1", "0"):
    os.environ["CHB_POOL_TAU"] = pool
    ctx = _lib.Context(0)
    ctx.set_samples(X)
    lab, its, ch, mind = ctx.fit_cluster(B, initial, perms, m, sweeps, want_min_dist=True)
    out[pool] = (lab, its, ch, mind)
    print("pool", pool, "sweeps", its, "changed", ch, ctx.fit_stats(), {k: ctx.counter(k) for k in ("pool_state", "pool_batches", "prefilter_overflow")}, flush=True)
    ctx.close()
a, b = out["1"], out["0"]
print("labels equal:", bool(np.array_equal(a[0], b[0])), "differing:", int((a[0] != b[0]).sum()), "min-dist max diff:",
      float(np.nanmax(np.abs(a[3] - b[3]))))
