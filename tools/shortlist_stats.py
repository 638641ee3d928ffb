"""Developer probe: shortlist lengths of the fp16 shortlist stage (base stage, last batch).
Run with CHB_PF_UPDATE=0 so that the update stage does not overwrite the candidate buffers."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import chbin_amd  # noqa: E402,F401
from chbin_amd import _lib, synth  # noqa: E402

N, D, B, m = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (100000, 136, 64, 5)))
S = 1 if D <= 136 else (5 if D == 140 else 10)
X, initial, true = synth.make_synthetic(N, D, B, S=S, seed=0)
perms = synth.draw_permutations(initial, 1, seed=0)
ctx = _lib.Context(0)
ctx.set_samples(X)
lab, _, _ = ctx.fit_cluster(B, initial, perms, m, 1)
st = ctx.fit_stats()
k_last = ctx.counter("last_batch_k")
s = ctx.counter("shortlist_sum_last_batch")
mx = ctx.counter("shortlist_max_last_batch")
print("N,D,B,m", N, D, B, m, "batches", st["batches"], "last batch K", k_last, "shortlist mean",
      s / (B * k_last), "max", mx, "overflow pairs (whole sweep)", ctx.counter("prefilter_overflow"))
