"""Turn the rocprofv3 --pmc passes of tools/collect_profiles.sh into the per-kernel table bench.py reads.

usage: python tools/traffic_from_pmc.py <dir with the pmc passes> <out prefix, e.g. profiles/r04> [--exclude <subdir>]
Writes <prefix>_pmc_hbm_counters.csv (every counter, per kernel) and <prefix>_traffic.json.

Per kernel (averages per launch):
  traffic_bytes_per_launch  = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / launches: what crossed the fabric.  FETCH_SIZE is
                              doubled as MI355X_MICROARCH.md prescribes for gfx950 (it reports half the bytes of wide
                              coalesced reads).
  l2_read_bytes_per_launch  = TCP_TCC_READ_REQ_sum * 128 / launches: what the CUs' vector L1s requested from the XCDs'
                              L2 (a request is one 128-B line: colsum_partial / absmax, which stream the whole matrix
                              once, show N * Dp * 8 / 128 requests).
  l2_hit_rate               = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
  valu_busy                 = SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES: share of a SIMD's time with a vector instruction in
                              issue (quad-cycles summed over waves = 4 cycles each, over 4 SIMDs per busy CU cycle)
  mfma_busy, mfma_coexec    = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES), COEXEC / MFMA_BUSY
  wait_any, wait_inst, active = shares of SQ_WAVE_CYCLES (parked at s_waitcnt / barrier; issue-stalled; issuing)
Fields whose pass is missing are null.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_stamp  # noqa: E402  (bench.py quotes the table only for the same kernel code (comments and layout aside))

src, prefix = sys.argv[1], sys.argv[2]
exclude = sys.argv[sys.argv.index("--exclude") + 1] if "--exclude" in sys.argv else None
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
    if exclude and (os.sep + exclude + os.sep) in f[len(src):]:
        continue
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("chb::(anonymous namespace)::", "").replace("void ", "")
        k = re.sub(r"\(.*$", "", k)
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
with open(prefix + "_pmc_hbm_counters.csv", "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["kernel", "Counter_Name", "total", "launches"])
    for k in sorted(tot):
        for c in sorted(tot[k]):
            w.writerow([k, c, tot[k][c], calls[k][c]])


def per_launch(k, c):
    return tot[k][c] / calls[k][c] if calls[k].get(c) else None


def ratio(a, b):
    return (a / b) if (a is not None and b) else None


kern = {}
for k in sorted(tot):
    if "FETCH_SIZE" not in tot[k] and "TCP_TCC_READ_REQ_sum" not in tot[k] and "SQ_WAVE_CYCLES" not in tot[k]:
        continue
    e = {"launches": max(calls[k].values())}
    if "FETCH_SIZE" in tot[k] and "WRITE_SIZE" in tot[k]:
        n = calls[k]["FETCH_SIZE"]
        e.update({"fetch_kb": tot[k]["FETCH_SIZE"], "write_kb": tot[k]["WRITE_SIZE"],
                  "traffic_bytes_per_launch": (2.0 * tot[k]["FETCH_SIZE"] + tot[k]["WRITE_SIZE"]) * 1024.0 / n})
    req = per_launch(k, "TCP_TCC_READ_REQ_sum")
    e["l2_read_bytes_per_launch"] = req * 128.0 if req is not None else None
    hit, miss = per_launch(k, "TCC_HIT_sum"), per_launch(k, "TCC_MISS_sum")
    e["l2_hit_rate"] = ratio(hit, (hit or 0) + (miss or 0))
    wc, cu = per_launch(k, "SQ_WAVE_CYCLES"), per_launch(k, "SQ_BUSY_CU_CYCLES")
    e["valu_busy"] = ratio(per_launch(k, "SQ_ACTIVE_INST_VALU"), cu)
    mb = per_launch(k, "SQ_VALU_MFMA_BUSY_CYCLES")
    e["mfma_busy"] = ratio(mb, 4.0 * cu if cu else None)
    e["mfma_coexec"] = ratio(per_launch(k, "SQ_VALU_MFMA_COEXEC_CYCLES"), mb)
    e["wait_any"] = ratio(per_launch(k, "SQ_WAIT_ANY"), wc)
    e["wait_inst"] = ratio(per_launch(k, "SQ_WAIT_INST_ANY"), wc)
    e["active"] = ratio(per_launch(k, "SQ_ACTIVE_INST_ANY"), wc)
    kern[k] = e
json.dump({"source": "rocprofv3 --pmc passes (each its own run, --kernel-trace only beside --pmc) on bench.py --steps 1 "
                     "--warmup 0 --cpu-sample 0 --no-e2e --no-extra: FETCH_SIZE / WRITE_SIZE (fabric side; FETCH_SIZE doubled "
                     "per MI355X_MICROARCH.md, Infinity-Cache hits included), TCP_TCC_READ_REQ_sum (L2 -> L1, 128-B requests), "
                     "TCC_HIT / MISS, SQ busy shares",
           "kernel_source_stamp": kernel_source_stamp(),
           "kernels": kern}, open(prefix + "_traffic.json", "w"), indent=1)
print("kernels:", ", ".join(kern))
