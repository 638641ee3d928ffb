"""Developer probe: per-kernel sums of rocprofv3 --pmc counter_collection CSVs.  usage: pmc_report.py <dir> [name filter]"""
import collections
import csv
import glob
import re
import sys

tot = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("chb::(anonymous namespace)::", "").replace("void ", ""))[:44]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k in sorted(tot):
    if flt and flt not in k:
        continue
    print(k)
    for c in sorted(tot[k]):
        print(f"    {c:32s} {tot[k][c]:16.0f}  / {n[k][c]} launches = {tot[k][c]/n[k][c]:14.0f}")
