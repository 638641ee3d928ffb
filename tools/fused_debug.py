"""Developer probe: fused selection path vs the list-based path (CHB_FUSED=0) vs the oracle on small fits."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import chbin_amd  # noqa: E402,F401
from chbin_amd import _lib, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def ctx_with(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return _lib.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


cases = [(600, 40, 6, 5, 3, 9e-3, 0.5, 10, 0), (2000, 136, 8, 5, 2, 1.5e-3, 0.0, None, 0),
         (900, 136, 8, 5, 3, 6e-3, 0.6, 8, 257), (20000, 136, 64, 5, 1, 1.5e-3, 0.0, None, 0)]
for (N, D, B, m, iters, sigma, mix, n_seed, batch) in cases:
    X, initial, _ = synth.make_synthetic(N, D, B, seed=N + B, sigma=sigma, mix=mix, n_seed=n_seed)
    perms = synth.draw_permutations(initial, iters, seed=0)
    res = {}
    for name, env in (("fused", {}), ("lists", {"CHB_FUSED": "0"})):
        c = ctx_with(env)
        c.set_samples(X)
        lab, its, ch, mind = c.fit_cluster(B, initial, perms, m, iters, batch=batch, want_min_dist=True)
        res[name] = (lab, its, ch, mind, c.fit_stats(), c.counter("fused_enabled"), c.counter("slow_pairs_last_round"),
                     c.counter("prefilter_overflow"))
        c.close()
    a, b = res["fused"], res["lists"]
    bad = np.flatnonzero(a[0] != b[0])
    print(f"case N={N} D={D} B={B} m={m}: fused rounds {a[4]['rounds']} lists rounds {b[4]['rounds']} "
          f"label mismatches {len(bad)} first {bad[:5]} slow(last round) {a[6]} overflow {a[7]}/{b[7]} fused_enabled {a[5]}/{b[5]}")
    md = np.abs(a[3] - b[3])
    md = md[np.isfinite(md)]
    print("   max |min_dist diff|", md.max() if len(md) else None, "changed per sweep", list(a[2]), list(b[2]))
    if N <= 2000:
        want, its_o, ch_o = O.fit_cluster(X, B, initial, perms, m, iters)
        print("   vs oracle: fused", int((a[0] != want).sum()), "lists", int((b[0] != want).sum()))
