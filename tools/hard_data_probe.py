"""Developer probe: overlapping bins at the benchmark size -- rounds per batch, sweeps, timing."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import chbin_amd  # noqa: E402,F401
from chbin_amd import _lib, synth  # noqa: E402

N, D, B, m = 100000, 136, 64, 5
mix = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 6e-3
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 0
X, initial, true = synth.make_synthetic(N, D, B, seed=0, sigma=sigma, mix=mix)
perms = synth.draw_permutations(initial, 10, seed=0)
ctx = _lib.Context(0)
ctx.set_samples(X)
ctx.fit_cluster(B, initial, perms[:1], m, 1, batch=batch)
t = time.perf_counter()
lab, its, ch = ctx.fit_cluster(B, initial, perms, m, 10, batch=batch)
dt = time.perf_counter() - t
st = ctx.fit_stats()
print(f"mix={mix} sigma={sigma} batch={batch}: sweeps={its} changed={list(ch)} acc={(lab == true).mean():.4f} "
      f"time={dt:.3f}s batches={st['batches']} rounds={st['rounds']} "
      f"evaluated/needed={st['hull_evaluated'] / st['hull_needed']:.3f} overflow={ctx.counter('prefilter_overflow')}")
