"""Developer check of the [short] lines of the tile-skipping build (CHB_SKIP_NEVER=32): is the tau the kernel derived from
sweep 0 really an upper bound of S x (the 5th nearest distance to the bin's initially labelled members)?"""
import sys, os, re, subprocess
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import chbin_amd
from chbin_amd import synth
N, D, B = 500000, 140, 128
X, initial, true = synth.make_synthetic(N, D, B, S=5, seed=0)
sizes = np.bincount(initial[initial >= 0], minlength=B)
n = 0
for line in open(sys.argv[1]):
    m = re.match(r"\[short\] bin (\d+) \((\d+) members, (\d+) tiles\) pos (\d+) sample (\d+) .*count (\d+) hits (\d+) .*hi_s1 (\S+) zn_lo (\S+) tau (\S+) S (\S+)", line)
    if not m: continue
    c, nb, nt, pos, smp, cnt, hits = (int(m.group(i)) for i in range(1, 8))
    hi, zn, tau, S = (float(m.group(i)) for i in range(8, 12))
    if nb != sizes[c]: continue          # (the bin has taken on members since the start)
    memb = np.flatnonzero(initial == c)
    d = np.sort(np.linalg.norm(X[memb] - X[smp], axis=1))
    mu = X[memb].mean(axis=0)
    print(f"bin {c} sample {smp}: count {cnt}; S*d5 = {S*d[4]:.3f}  S*d4 = {S*d[3]:.3f}  S*d6 = {S*d[5]:.3f}   kernel tau = {tau:.3f}  hi_s1 = {hi:.3f};  S*|x - mu| = {S*np.linalg.norm(X[smp]-mu):.3f} zn_lo = {zn:.3f}")
    n += 1
    if n >= 12: break
