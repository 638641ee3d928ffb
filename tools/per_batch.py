"""Developer probe: per-launch durations (us) of the kernels matching a name in a rocprofv3 kernel trace CSV.
usage: per_batch.py <kernel_trace.csv> [--first N] name ..."""
import csv
import sys
args = sys.argv[1:]
path = args.pop(0)
first = 0
if args and args[0] == "--first":
    first = int(args[1]); args = args[2:]
rows = list(csv.DictReader(open(path)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
for name in args:
    d = [(e - s) / 1e3 for s, e, k in ev if name in k]
    sel = d[:first] if first else d[-14:]
    print(name, len(d), [round(x) for x in sel], 'sum', round(sum(sel)))
