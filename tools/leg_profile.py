"""Developer probe: per-kernel time of one sweep on a generator setting (the bench's extra legs), with and without the pools.
usage: python tools/leg_profile.py mix sigma [m]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chbin_amd  # noqa: E402,F401
from chbin_amd import _lib, synth  # noqa: E402

mix, sigma = float(sys.argv[1]), float(sys.argv[2])
m = int(sys.argv[3]) if len(sys.argv) > 3 else 5
N, D, B = 100_000, 136, 64
X, initial, true = synth.make_synthetic(N, D, B, S=1, seed=0, mix=mix, sigma=sigma)
perms = synth.draw_permutations(initial, 1, seed=0)
names = ("prefilter", "prefilter_retry", "prefilter_update", "topm_fallback", "hull_qp", "slow_path", "argmin", "bucket", "pool", "query_norms", "fit_start")
for pool in ("1", "0"):
    os.environ["CHB_POOL_TAU"] = pool
    ctx = _lib.Context(0)
    ctx.set_samples(X)
    ctx.fit_cluster(B, initial, perms, m, 1)
    ctx.profile_reset(); ctx.profile_enable(1)
    import time
    t0 = time.perf_counter()
    ctx.fit_cluster(B, initial, perms, m, 1)
    dt = time.perf_counter() - t0
    ctx.profile_enable(0)
    pr = {k: ctx.profile_get(k) for k in names}
    print("pool", pool, "ms %.1f" % (dt * 1e3), {k: ctx.counter(k) for k in ("pool_state", "pool_batches", "pool_candidates", "pool_pairs", "prefilter_overflow", "segment_batches")}, ctx.fit_stats())
    print("    ", ", ".join("%s %.2f (%d)" % (k, v["ms"], v["launches"]) for k, v in sorted(pr.items(), key=lambda kv: -kv[1]["ms"]) if v["launches"]))
    ctx.close()
