"""Per-step kernel breakdown from a rocprofv3 kernel trace of bench.py.
usage: python tools/trace_steps.py <kernel_trace.csv> [n_last_steps] [top]"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 5
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'attn_fwd_kernel' in r['Kernel_Name']]
starts = idx[::12]
print(len(starts), 'steps seen; step-to-step ms:', [round((int(rows[starts[i+1]]['Start_Timestamp']) - int(rows[starts[i]]['Start_Timestamp'])) / 1e6, 2) for i in range(len(starts) - 1)][-12:])
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0        # steps to skip at the end (eager kernel-timing steps)
a, b = starts[-(nlast + 1 + skip)], starts[-(1 + skip)]
seg = rows[a:b]
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
span = int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])
print(f'{nlast} steps: kernels/step {len(seg)/nlast:.0f}  busy {busy/nlast/1e6:.2f} ms/step  span {span/nlast/1e6:.2f} ms/step')
d = defaultdict(lambda: [0, 0])
for r in seg:
    k = r['Kernel_Name'][:100]
    d[k][0] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    d[k][1] += 1
for k, v in sorted(d.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{k:100s} {v[0]/nlast/1e6:7.3f} ms/step  n={v[1]/nlast:6.1f}  avg_us={v[0]/v[1]/1e3:8.1f}")
