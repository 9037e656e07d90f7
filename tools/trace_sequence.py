"""One step's kernel sequence from a rocprofv3 kernel trace (csv): index, duration, gap to the previous kernel, name.
usage: python tools/trace_sequence.py <kernel_trace.csv> [marker substring = adamw_flat]   (the step between the last two markers)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "adamw_flat"
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
busy = 0.0
for i in range(a + 1, b + 1):
    r = rows[i]
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:90]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    gap = (int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3
    busy += d
    print(f"{i - a:4d} {d:8.1f} us  gap {gap:6.1f}  grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):6d}  {n}")
span = (int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3
print(f"kernels {b - a}  busy {busy:.1f} us  span {span:.1f} us")
