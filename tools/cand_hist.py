"""Developer probe: distribution of shortlist lengths (base stage) of the last batch of sweep 1.
Needs the developer library (make -C ch-bin_amd/csrc DEV=1; CHBIN_LIB=ch-bin_amd/libchbin_hip_dev.so)
with CHB_PF_UPDATE=0 so that the update stage does not overwrite the candidate buffers."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import chbin_amd  # noqa: E402,F401
from chbin_amd import _lib, synth  # noqa: E402

N, D, B, m = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (100000, 136, 64, 5)))
mix = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
sigma = float(sys.argv[6]) if len(sys.argv) > 6 else 1.5e-3
S = 1 if D <= 136 else (5 if D == 140 else 10)
X, initial, true = synth.make_synthetic(N, D, B, S=S, seed=0, mix=mix, sigma=sigma)
perms = synth.draw_permutations(initial, 1, seed=0)
ctx = _lib.Context(0)
ctx.set_samples(X)
ctx.fit_cluster(B, initial, perms, m, 1)
k_last = ctx.counter("last_batch_k")
tot = B * k_last
print("N,D,B,m,mix,sigma", N, D, B, m, mix, sigma, "last batch K", k_last, "mean",
      ctx.counter("shortlist_sum_last_batch") / tot, "max", ctx.counter("shortlist_max_last_batch"))
for lim in (m - 1, m, m + 1, m + 2, m + 3, 8, 10, 12, 16, 24, 32, 64):
    print(f"  <= {lim:3d}: {ctx.counter('shortlist_le%d_last_batch' % lim) / tot:.5f}")
