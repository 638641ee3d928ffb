"""Developer probe: idle time between kernels from a rocprofv3 --kernel-trace CSV.

usage: python tools/gap_report.py <..._kernel_trace.csv> [skip_first_n_kernels]
Prints busy / idle totals of the stream and the largest contributors to idle time by (previous kernel
-> next kernel) pair."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))[skip:]


def short(k):
    k = k.replace("chb::(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", k)[:48]


busy = sum(e - s for s, e, _ in ev)
span = ev[-1][1] - ev[0][0]
gaps = collections.defaultdict(lambda: [0, 0])
for (s0, e0, k0), (s1, e1, k1) in zip(ev, ev[1:]):
    g = max(0, s1 - e0)
    key = (short(k0), short(k1))
    gaps[key][0] += g
    gaps[key][1] += 1
print(f"kernels {len(ev)}  span {span/1e6:.3f} ms  busy {busy/1e6:.3f} ms  idle {(span-busy)/1e6:.3f} ms")
for key, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"{g/1e3:9.1f} us total {g/1e3/n:7.2f} us avg x{n:5d}   {key[0]} -> {key[1]}")
per = collections.defaultdict(lambda: [0, 0])
for s, e, k in ev:
    per[short(k)][0] += e - s
    per[short(k)][1] += 1
print("--- busy by kernel")
for k, (t, n) in sorted(per.items(), key=lambda kv: -kv[1][0])[:30]:
    print(f"{t/1e3:10.1f} us  x{n:5d}  avg {t/1e3/n:8.2f}  {k}")
