"""Per-kernel averages of the counters of a `rocprofv3 --pmc ...` run (counter_collection.csv), one line per kernel name.
usage: python tools/pmc_kernels.py <counter_collection.csv> [substring filter]"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if len(sys.argv) > 2 and sys.argv[2] not in n:
        continue
    a = acc[n][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"])
    a[1] += 1
for n, cs in sorted(acc.items()):
    print(n[:70], " ".join(f"{k}={v[0] / v[1]:.4g}" for k, v in sorted(cs.items())), f"n={max(v[1] for v in cs.values())}")
