"""Developer soak of the tile-skipping shortlist builds: medium-size fits on data with several coverage columns (so that
tiles really are skipped) against the CPU oracle's WHOLE fit, under the developer library's shortlist validation.
usage: CHBIN_LIB=.../libchbin_hip_dev.so CHB_SL_VALIDATE=1 python tools/soak_skip.py [n_cases] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import chbin_amd  # noqa: E402,F401
from chbin_amd import _lib, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = _lib.default_context()
bad = 0
for t in range(n_cases):
    D = int(rng.choice([140, 141, 143, 146, 150]))
    S = 5 if D < 146 else 10
    N = int(rng.integers(6000, 14000)); B = int(rng.integers(6, 20)); m = int(rng.choice([3, 5, 5, 8]))
    n_seed = int(rng.integers(40, 200)); iters = 2
    X, initial, _ = synth.make_synthetic(N, D, B, S=S, seed=int(rng.integers(1 << 30)), sigma=float(rng.choice([1.5e-3, 4e-3])),
                                         mix=float(rng.choice([0.0, 0.3])), n_seed=n_seed)
    perms = synth.draw_permutations(initial, iters, seed=int(rng.integers(1 << 30)))
    t0 = time.time()
    want, its_o, ch_o = O.fit_cluster(X, B, initial, perms, m, iters)
    t1 = time.time()
    ctx.set_samples(X)
    got, its, ch = ctx.fit_cluster(B, initial, perms, m, iters, batch=int(rng.choice([0, 2048])))
    ok = its == its_o and np.array_equal(ch, ch_o) and np.array_equal(got, want)
    bad += not ok
    print(f"[{t}] N={N} D={D} B={B} m={m} seeds={n_seed}: {'ok' if ok else 'MISMATCH'} (oracle {t1 - t0:.0f} s; skip state "
          f"{ctx.counter('tile_skip_state')}, never loaded {ctx.counter('tile_unloaded')} of {ctx.counter('tile_seen') + ctx.counter('tile_unloaded')} sampled)",
          flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
