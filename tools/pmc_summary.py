"""Mean of each hardware counter per kernel over the launches in a rocprofv3 --pmc output directory (development aid)."""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            if pat and pat not in k:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k[:110])
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
