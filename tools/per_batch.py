"""Developer probe: per-launch durations (us) of the kernels matching a name in a rocprofv3 kernel trace CSV."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
for name in sys.argv[2:]:
    d = [(e - s) / 1e3 for s, e, k in ev if name in k]
    print(name, len(d), [round(x) for x in d[-14:]], 'sum', round(sum(d[-14:])))
