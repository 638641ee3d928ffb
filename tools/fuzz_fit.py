"""Developer soak: random small configurations, HIP fit_cluster vs the CPU oracle (labels, sweep
counts, per-sweep change counts must be identical).  usage: python tools/fuzz_fit.py [n_cases] [seed] [big|m16]   (big: bins of > 512 members, few bins; m16: the fused
16-lane kernel, 6 <= m <= 16; manybins: 65 .. 400 bins of a handful of members -- more than one 64-bin tile of the
per-fit query-norm table; wide: rows of 158 .. 573 columns -- the wide shortlist builds of round 5 and the fused kernels on them)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import chbin_amd  # noqa: E402,F401
from chbin_amd import _lib, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
big = len(sys.argv) > 3 and sys.argv[3] == "big"
m16 = len(sys.argv) > 3 and sys.argv[3] == "m16"
manybins = len(sys.argv) > 3 and sys.argv[3] == "manybins"
wide = len(sys.argv) > 3 and sys.argv[3] == "wide"
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = _lib.default_context()
bad = 0
unloaded = seen = 0
for t in range(n_cases):
    N = int(rng.integers(200, 1800))
    D = int(rng.choice([8, 24, 40, 100, 136, 137, 140, 143, 144, 145, 146, 160, 161, 200]))
    B = int(rng.integers(1, 24))
    m = int(rng.choice([1, 2, 3, 5, 5, 5, 8, 9, 15, 16]))
    S = 1 if D < 140 else (5 if D < 146 else 10)
    iters = int(rng.integers(1, 6))
    batch = int(rng.choice([0, 1, 7, 64, 100, 257, 1000, 4096]))
    sigma = float(rng.choice([1.5e-3, 4e-3, 9e-3]))
    mix = float(rng.choice([0.0, 0.3, 0.6, 0.9]))
    n_seed = int(rng.integers(1, 12))
    if big:
        N = int(rng.integers(3000, 7000)); B = int(rng.integers(2, 7)); m = int(rng.choice([3, 5, 5, 8]))
        iters = int(rng.integers(1, 4)); batch = int(rng.choice([0, 512, 2048])); n_seed = int(rng.integers(5, 40))
        D = int(rng.choice([100, 136, 140, 146]))
        S = 1 if D < 140 else (5 if D < 146 else 10)
    if m16:
        N = int(rng.integers(200, 1100)); m = int(rng.integers(6, 17)); D = int(rng.choice([24, 40, 100, 136, 140, 146, 160]))
        S = 1 if D < 140 else (5 if D < 146 else 10)
        iters = int(rng.integers(1, 4)); n_seed = int(rng.integers(1, 24))
    if manybins:
        N = int(rng.integers(2000, 5000)); B = int(rng.integers(65, 400)); m = int(rng.choice([1, 3, 5, 5]))
        D = int(rng.choice([100, 136, 140, 146])); S = 1 if D < 140 else (5 if D < 146 else 10)
        iters = int(rng.integers(1, 3)); batch = int(rng.choice([0, 300, 1000])); n_seed = int(rng.integers(1, 4))
    if wide:
        N = int(rng.integers(200, 1300)); D = int(rng.choice([158, 200, 285, 286, 300, 429, 430, 528, 573, 574]))
        S = int(rng.choice([1, 1, 5, 10])); iters = int(rng.integers(1, 4))
    metric = str(rng.choice(["convex", "convex", "convex", "affine"]))
    if m > D or D < 24 or (D < 40 and m > 8):   # (round 4, case `120 91 m16` #101: 14 vertices in D = 24 straddle orth's cutoff too)
        # m > D: the affine hull of > D generic points is the whole space, every distance is rounding
        # noise.  Small D: the reference formula (scipy.linalg.orth with its default cutoff eps * max(m, D),
        # restated in the oracle) can admit the pure-noise direction of the centred vertex matrix
        # (DESIGN.md section 2), so there is no stable reference value to compare with.
        metric = "convex"
    X, initial, _ = synth.make_synthetic(N, D, B, S=min(S, max(D - 4, 1)), seed=int(rng.integers(1 << 30)), sigma=sigma,
                                         mix=mix, n_seed=n_seed)
    if rng.random() < 0.2 and m <= 8:   # (the oracle's nearest-PD / GI restatement crawls on rescaled 15-vertex hulls)
        X = X * float(10.0 ** rng.integers(-6, 7))
    perms = synth.draw_permutations(initial, iters, seed=int(rng.integers(1 << 30)))
    print(f"[{t:3d}] N={N} D={D} B={B} m={m} iters={iters} batch={batch} ...", flush=True)
    want, its_o, ch_o = O.fit_cluster(X, B, initial, perms, m, iters, metric=metric)
    ctx.set_metric(metric)
    ctx.set_samples(X)
    got, its, ch = ctx.fit_cluster(B, initial, perms, m, iters, batch=batch)
    ok = its == its_o and np.array_equal(ch, ch_o) and np.array_equal(got, want)
    bad += not ok
    unloaded += ctx.counter("tile_unloaded"); seen += ctx.counter("tile_seen")
    print(f"[{t:3d}] N={N} D={D} B={B} m={m} iters={iters} batch={batch} sigma={sigma} mix={mix} seeds={n_seed} "
          f"{metric}: {'ok' if ok else 'MISMATCH'} (sweeps {its}/{its_o}, diff labels {int((got != want).sum())})", flush=True)
ctx.set_metric("convex")
print("mismatches:", bad, " tiles never loaded / met (sampled):", unloaded, seen)
sys.exit(1 if bad else 0)
