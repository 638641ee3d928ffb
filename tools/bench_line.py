"""Developer helper: one-line summary of a bench.py JSON line.  usage: bench_line.py <file> [label]"""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2] if len(sys.argv) > 2 else sys.argv[1], round(d["ms_per_step"], 3),
      [(k["kernel"], round(k["ms_per_step"], 3)) for k in (d.get("kernels") or [])[:4]], d.get("fit_stats_last_call"))
