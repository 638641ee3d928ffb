"""Developer probe: sweep-1 time, speculation rounds, accuracy and bin-size skew of the synthetic generator at several
overlap settings (mix = pull of the bins towards a common profile, sigma = within-bin spread).
usage: python tools/overlap_scan.py [N D B m]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import chbin_amd  # noqa: E402,F401
from chbin_amd import _lib, synth  # noqa: E402

N, D, B, m = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (100000, 136, 64, 5)))
ctx = _lib.Context(0)
for mix, sigma in ((0.0, 1.5e-3), (0.3, 3e-3), (0.5, 3e-3), (0.3, 4.5e-3), (0.7, 1.5e-3), (0.7, 3e-3), (0.5, 4.5e-3), (0.5, 6e-3)):
    X, initial, true = synth.make_synthetic(N, D, B, S=1, seed=0, mix=mix, sigma=sigma)
    perms = synth.draw_permutations(initial, 1, seed=0)
    ctx.set_samples(X)
    ctx.fit_cluster(B, initial, perms, m, 1)
    t = time.perf_counter()
    lab, _, _ = ctx.fit_cluster(B, initial, perms, m, 1)
    dt = time.perf_counter() - t
    st = ctx.fit_stats()
    h = np.sort(np.bincount(lab[lab >= 0], minlength=B))[::-1]
    print(f"mix={mix} sigma={sigma}: {dt*1e3:7.2f} ms/sweep, rounds/batch {st['rounds']/max(st['batches'],1):.2f}, "
          f"evaluated/needed {st['hull_evaluated']/max(st['hull_needed'],1):.2f}, accuracy {(lab == true).mean():.4f}, "
          f"largest bins {h[:3]}, overflows {ctx.counter('prefilter_overflow')}", flush=True)
