"""Developer probe: many small bins -- the two-stage selection against the brute-force selection
(CHB_PREFILTER=0) on the same data; labels, sweep counts and change counts must be identical."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import chbin_amd  # noqa: E402,F401
from chbin_amd import _lib, synth  # noqa: E402

N, D, B, m = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (20000, 136, 500, 5)))
S = 1 if D <= 136 else (5 if D == 140 else 10)
X, initial, true = synth.make_synthetic(N, D, B, S=S, seed=3, sigma=3e-3, mix=0.3, n_seed=3)
perms = synth.draw_permutations(initial, 3, seed=0)
a = _lib.Context(0)
a.set_samples(X)
la, ia, ca = a.fit_cluster(B, initial, perms, m, 3)
print("two-stage:", ia, list(ca), "overflow", a.counter("prefilter_overflow"), "enabled", a.counter("prefilter_enabled"))
a.close()
os.environ["CHB_PREFILTER"] = "0"
b = _lib.Context(0)
b.set_samples(X)
lb, ib, cb = b.fit_cluster(B, initial, perms, m, 3)
print("brute    :", ib, list(cb), "enabled", b.counter("prefilter_enabled"))
ok = ia == ib and np.array_equal(ca, cb) and np.array_equal(la, lb)
print("identical:", ok)
sys.exit(0 if ok else 1)
