"""Developer probe: throughput of the k-mer kernel (bases/s, kernel only and incl. host copies)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import chbin_amd  # noqa: E402,F401
from chbin_amd import _lib  # noqa: E402

n, mean_len, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) >= 4 else (20000, 10000, 4)))
rng = np.random.default_rng(0)
lengths = rng.integers(mean_len // 2, mean_len * 3 // 2, size=n)
blob = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(lengths.sum())).tobytes()
off = np.concatenate([[0], np.cumsum(lengths)])
seqs = [blob[off[i]:off[i + 1]] for i in range(n)]
ctx = _lib.default_context()
ctx.kmer_frequencies(seqs[:10], k)
ctx.profile_enable(True)
ctx.profile_reset()
t0 = time.perf_counter()
f = ctx.kmer_frequencies(seqs, k)
dt = time.perf_counter() - t0
p = ctx.profile_get("kmer_count")
print(f"contigs {n} bases {len(blob):.3e} k {k}: kernel {p['ms']:.3f} ms = {len(blob) / p['ms'] / 1e6:.2f} Gbase/s; "
      f"whole call (join + H2D + kernel + D2H) {dt * 1e3:.1f} ms; row sums ok {np.allclose(f.sum(1), 1.0)}")
