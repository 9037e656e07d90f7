"""SQ_VALU_MFMA_BUSY_CYCLES of the window-attention kernels -> profiles/r02_pmc_window_attention_mfma.json (quoted by bench.py).

usage: python tools/pmc_mfma_summary.py <counter_collection.csv> <kernel_stats.csv or kernel_trace.csv> <out.json> [extra csv pairs...]

mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the SIMDs) / (kernel time x 1024 SIMDs x clock): the share of the
matrix pipes' cycles in which an MFMA executes.  It counts the 64-row padded tiles (49 real tokens), so it is >= the
algorithmic fraction bench.py reports.  The clock is not observable from these passes: the fraction is given at the 2.4
GHz maximum (a lower bound; under MFMA load the chip holds 1.9-2.2 GHz) with the raw numbers beside it."""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

NAMES = {"win_fused_fwd_kernel<false>": "pswin_win_attn_fused_fwd (inference)", "win_fused_fwd_kernel<true>": "pswin_win_attn_fused_fwd (training saves)",
         "attn_fwd_kernel": "pswin_attn_fwd", "attn_bwd_pair_kernel": "pswin_attn_bwd"}
out = {"lib_digest": bench._lib_digest(), "counter": "SQ_VALU_MFMA_BUSY_CYCLES (rocprofv3 --pmc, own pass)", "kernels": {}}
args = sys.argv[1:]
outp = args[2]
pairs = [(args[0], args[1])] + [(args[i], args[i + 1]) for i in range(3, len(args) - 1, 2)]
for pmc, trace in pairs:
    busy, dur = defaultdict(list), defaultdict(list)
    for r in csv.DictReader(open(pmc)):
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            for k, n in NAMES.items():
                if k in r["Kernel_Name"]:
                    busy[n].append(float(r["Counter_Value"]))
    for r in csv.DictReader(open(trace)):
        name = r.get("Kernel_Name") or r.get("Name")
        for k, n in NAMES.items():
            if k in name:
                if "AverageNs" in r:
                    dur[n].append(float(r["AverageNs"]))
                else:
                    dur[n].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for n in busy:
        if not dur[n]:
            continue
        b, d = sum(busy[n]) / len(busy[n]), sum(dur[n]) / len(dur[n])
        out["kernels"][n] = {"mfma_busy_cycles_per_launch": b, "avg_launch_ns": d, "mfma_busy_at_2.4GHz": round(b / (d * 1e-9 * 2.4e9 * 1024), 4),
                             "mfma_busy_at_2.0GHz": round(b / (d * 1e-9 * 2.0e9 * 1024), 4)}
json.dump(out, open(outp, "w"), indent=1)
print(json.dumps(out, indent=1))
