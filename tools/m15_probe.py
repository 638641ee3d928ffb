"""Developer probe: m = 15 sweep timing, 16-lane solver iteration statistics (developer library only), shortlist
width histogram, parity against the oracle on a prefix.
usage: python tools/m15_probe.py [m], with CHBIN_LIB pointing at a variant built by
  tools/build_variant.sh stat "-DCHB_DEV_QP16_STAT" qp_kernels   (solver iteration statistics; distorts timings) or
  tools/build_variant.sh clk "-DCHB_DEV_CLK" qp_kernels           (cycle stamps of the fused 16-lane kernel)"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import chbin_amd  # noqa: E402,F401
from chbin_amd import _lib, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

N, D, B, m = 100000, 136, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 15
X, initial, true = synth.make_synthetic(N, D, B, seed=0)
perms = synth.draw_permutations(initial, 3, seed=0)
ctx = _lib.Context(0)
ctx.set_samples(X)
lab, _, _ = ctx.fit_cluster(B, initial, perms[:1], m, 1)
lib = ctypes.CDLL(_lib.LIB_PATH)
have = hasattr(lib, "chb_dev_qp16_stats")
st = (ctypes.c_ulonglong * 16)()
if have:
    lib.chb_dev_qp16_stats(st, 1)
t = time.perf_counter()
for _ in range(3):
    lab, _, _ = ctx.fit_cluster(B, initial, perms[:1], m, 1)
dt = (time.perf_counter() - t) / 3
print(f"last batch K={ctx.counter('last_batch_k')} x B={B} pairs; m={m}: {dt*1e3:.2f} ms/sweep, slow pairs (last round) {ctx.counter('slow_pairs_last_round')}, overflow {ctx.counter('prefilter_overflow')}")
if have:
    lib.chb_dev_qp16_stats(st, 0)
    p = max(st[0], 1)
    if st[0]:
        print(f"solver: {st[0]} problems, {st[1]/p:.2f} major iterations, {st[2]/p:.2f} removals, final support {st[3]/p:.2f}, "
          f"{st[4]/p:.3f} small-pivot refinements per problem")
if hasattr(lib, "chb_dev_qp16_clk"):
    ck = (ctypes.c_ulonglong * 8)()
    lib.chb_dev_qp16_clk(ck)
    w = max(ck[6], 1)
    print(f"fused 16-lane kernel, last launch, cycles per wavefront that reaches the end ({ck[6]}): sweeps {ck[0]/w:.0f}, "
          f"selection {ck[1]/w:.0f}, slow list + solver {ck[2]/w:.0f}; per one-tile pair ({ck[5]}): ids {ck[3]/max(ck[5],1):.0f}, "
          f"sweep + tile store {ck[4]/max(ck[5],1):.0f}; first-to-last wavefront start {ck[7]} cycles")
for n in (14, 15, 16, 17, 18, 19, 20, 22, 24, 28, 32, 48):
    try:
        print(f"  shortlist <= {n}: {ctx.counter('shortlist_le%d_last_batch' % n)}", end="")
    except Exception:
        pass
print()
ns = 40
lab_o, _ = O.sweep(X, B, initial, perms[0][:ns], m)
print("prefix parity:", np.array_equal(lab_o[perms[0][:ns]], lab[perms[0][:ns]]))
