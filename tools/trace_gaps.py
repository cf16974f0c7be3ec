"""Gaps between consecutive kernels of a rocprofv3 --kernel-trace csv (development aid): where an iteration's time
goes besides the kernels themselves."""
import csv, sys
from collections import defaultdict
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void bz::", "").split("(")[0][:60]))
rows.sort()
rows = rows[len(rows) // 2:]          # steady state
dur = defaultdict(list); gap = defaultdict(list)
for i, (s, e, k) in enumerate(rows):
    dur[k].append(e - s)
    if i:
        gap[(rows[i - 1][2], k)].append(s - rows[i - 1][1])
for k, v in dur.items():
    print(f"kernel {k:62s} n={len(v):5d} avg={sum(v)/len(v)/1e3:9.2f} us")
for k, v in sorted(gap.items(), key=lambda kv: -len(kv[1]))[:8]:
    print(f"gap {k[0][:40]:40s} -> {k[1][:40]:40s} n={len(v):5d} avg={sum(v)/len(v)/1e3:8.2f} us")
