import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import chbin_amd
from chbin_amd import _lib, synth
N, D, B = (int(x) for x in sys.argv[1:4])
m = int(sys.argv[4]) if len(sys.argv) > 4 else 5
S = 1 if D <= 136 else (5 if D == 140 else 10)
X, initial, true = synth.make_synthetic(N, D, B, S=S, seed=0)
perms = synth.draw_permutations(initial, 2, seed=0)
ctx = _lib.Context(0)
ctx.set_samples(X)
lab, its, ch = ctx.fit_cluster(B, initial, perms, m, 2)
print("skip state", ctx.counter("tile_skip_state"), "skipped", ctx.counter("tile_skipped"), "seen", ctx.counter("tile_seen"), "unloaded", ctx.counter("tile_unloaded"),
      "overflow", ctx.counter("prefilter_overflow"), "acc", float((lab == true).mean()))
