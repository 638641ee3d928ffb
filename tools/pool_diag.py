"""Developer probe: one sweep of a synthetic configuration, then the threshold pools' counters.
usage: python tools/pool_diag.py N D B [m] [sweeps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chbin_amd  # noqa: E402,F401
from chbin_amd import _lib, synth  # noqa: E402

N, D, B = (int(v) for v in sys.argv[1:4])
m = int(sys.argv[4]) if len(sys.argv) > 4 else 5
sweeps = int(sys.argv[5]) if len(sys.argv) > 5 else 1
S = 1 if D <= 136 else (5 if D == 140 else 10)
X, initial, true = synth.make_synthetic(N, D, B, S=S, seed=0)
perms = synth.draw_permutations(initial, sweeps, seed=0)
ctx = _lib.Context(0)
ctx.set_samples(X)
for rep in range(2):
    t0 = time.perf_counter()
    lab, its, ch = ctx.fit_cluster(B, initial, perms, m, sweeps)
    dt = time.perf_counter() - t0
    st = ctx.fit_stats()
    print({k: ctx.counter(k) for k in ("pool_state", "pool_batches", "pool_candidates", "pool_pairs", "prefilter_overflow",
                                       "tile_skip_state", "lookahead_batches", "lookahead_failed")},
          st, "seconds %.3f" % dt, "accuracy %.4f" % float((lab == true).mean()), flush=True)
