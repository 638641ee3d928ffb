"""Developer probe: per-kernel digest of the collect_pmc.sh passes (ratios that say what a kernel waits on).
usage: pmc_digest.py gpurun_out/<tag> <cfg prefix, e.g. cfg2> [kernel substring ...]"""
import collections
import csv
import glob
import re
import sys

root, cfg = sys.argv[1], sys.argv[2]
filt = sys.argv[3:] or ["shortlist_kernel", "hull_select"]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
dur = collections.defaultdict(float); ndur = collections.defaultdict(int)
def kname(s):
    return re.sub(r"\(.*$", "", s.replace("chb::(anonymous namespace)::", "").replace("void ", ""))
for f in glob.glob(f"{root}/{cfg}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = kname(r["Kernel_Name"])
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for f in glob.glob(f"{root}/{cfg}_sqA/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = kname(r["Kernel_Name"])
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3; ndur[k] += 1
for k in sorted(tot):
    if not any(s in k for s in filt):
        continue
    c = {x: tot[k][x] / max(n[k][x], 1) for x in tot[k]}
    L = max(n[k].values())
    print(f"== {k}   launches {L}" + (f"   avg {dur[k]/ndur[k]:.1f} us (under --pmc)" if ndur[k] else ""))
    g = c.get
    wc = g("SQ_WAVE_CYCLES")
    if wc:
        print(f"   wave-cycles {wc:.3g}: WAIT_ANY {g('SQ_WAIT_ANY',0)/wc:.1%}  WAIT_INST_ANY {g('SQ_WAIT_INST_ANY',0)/wc:.1%}  "
              f"ACTIVE_INST_ANY {g('SQ_ACTIVE_INST_ANY',0)/wc:.1%} (VALU {g('SQ_ACTIVE_INST_VALU',0)/wc:.1%}, LDS {g('SQ_ACTIVE_INST_LDS',0)/wc:.1%})")
        busy = g("SQ_BUSY_CYCLES", 0)
        print(f"   SQ_BUSY_CYCLES {busy:.3g}  waves-in-flight avg = wave-cycles/busy = {wc/max(busy,1):.2f}")
    mb = g("SQ_VALU_MFMA_BUSY_CYCLES")
    if mb is not None and g("SQ_BUSY_CYCLES"):
        # MFMA_BUSY counts cycles per SIMD summed; BUSY_CYCLES quad-cycles per SE(?): report raw ratio + coexec share
        print(f"   MFMA_BUSY {mb:.3g}  COEXEC {g('SQ_VALU_MFMA_COEXEC_CYCLES',0):.3g}  coexec/mfma_busy {g('SQ_VALU_MFMA_COEXEC_CYCLES',0)/max(mb,1):.1%}  INSTS_MFMA {g('SQ_INSTS_MFMA',0):.3g}")
    if g("SQ_INSTS_VALU"):
        print(f"   insts/launch: VALU {g('SQ_INSTS_VALU',0):.3g} SALU {g('SQ_INSTS_SALU',0):.3g} LDS {g('SQ_INSTS_LDS',0):.3g} VMEM {g('SQ_INSTS_VMEM',0):.3g} "
              f"| LDS bank conflict {g('SQ_LDS_BANK_CONFLICT',0)/max(g('SQ_LDS_IDX_ACTIVE',1),1):.1%} of LDS-active")
    if g("SQ_INSTS_VMEM_RD") is not None:
        print(f"   VMEM_RD insts {g('SQ_INSTS_VMEM_RD',0):.3g}  ACTIVE_INST_VMEM {g('SQ_ACTIVE_INST_VMEM',0):.3g}  INST_LEVEL_VMEM {g('SQ_INST_LEVEL_VMEM',0):.3g}  "
              f"FMA_F64 {g('SQ_INSTS_VALU_FMA_F64',0):.3g}  WAVES {g('SQ_WAVES',0):.3g}  BUSY_CU_CYCLES {g('SQ_BUSY_CU_CYCLES',0):.3g}")
    if g("TCP_TCC_READ_REQ_sum") is not None:
        print(f"   TCP: TCC_READ_REQ {g('TCP_TCC_READ_REQ_sum',0):.4g}  TOTAL_CACHE_ACCESSES {g('TCP_TOTAL_CACHE_ACCESSES_sum',0):.4g}  TOTAL_READ {g('TCP_TOTAL_READ_sum',0):.4g}  "
              f"TA_DATA_STALL {g('TCP_TCP_TA_DATA_STALL_CYCLES_sum',0):.4g}")
    if g("TCC_REQ_sum") is not None:
        h, m_ = g("TCC_HIT_sum", 0), g("TCC_MISS_sum", 0)
        print(f"   TCC: REQ {g('TCC_REQ_sum',0):.4g} READ {g('TCC_READ_sum',0):.4g} HIT {h:.4g} MISS {m_:.4g}  hit rate {h/max(h+m_,1):.1%}")
