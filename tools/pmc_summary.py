"""Sum rocprofv3 --pmc counter_collection csv per kernel (developer tool).
usage: python tools/pmc_summary.py <dir> [kernel-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("chb::(anonymous namespace)::", "").split("(")[0]
        if sub and sub not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[(k, r["Counter_Name"])] += 1
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {v:16.0f}  ({calls[(k, c)]} rows)")
