"""Summarise a rocprofv3 --kernel-trace --stats CSV: python tools/prof_summary.py <kernel_stats.csv> <steps> [top]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total GPU kernel time {tot/1e6:.1f} ms = {tot/1e6/steps:.2f} ms/step over {steps:.0f} steps")
for r in rows[:top]:
    print(f"{r['Name'][:100]:100s} n/step={float(r['Calls'])/steps:6.1f} ms/step={float(r['TotalDurationNs'])/1e6/steps:7.3f} avg_us={float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):5.1f}%")
