"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv: python tools/pmc_rows.py <csv> [kernel substring]"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if len(sys.argv) > 2 and sys.argv[2] not in r["Kernel_Name"]:
        continue
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} launches {len(v):3d}  mean {sum(v) / len(v):16.1f}  last {v[-1]:16.1f}")
