"""SQ_VALU_MFMA_BUSY_CYCLES of the window-attention kernels -> profiles/r03_pmc_window_attention_mfma.json (quoted by bench.py).

usage: python tools/pmc_mfma_summary.py <counter_collection.csv> <kernel_stats.csv or kernel_trace.csv> <out.json> [extra csv pairs...]

mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the SIMDs) / (kernel time x 1024 SIMDs x clock): the share of the
matrix pipes' cycles in which an MFMA executes.  It counts the 64-row padded tiles (49 real tokens), so it is >= the
algorithmic fraction bench.py reports.  The clock is not observable from these passes: the fraction is given at the 2.4
GHz maximum (a lower bound) and, for the kernels whose in-kernel clock was measured (s_memtime / s_memrealtime stamps,
tools/probe/clock_probe.hip -> profiles/r03_fused_window_clock_probe.txt), at that clock."""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

NAMES = {"win_fused_fwd_kernel<false>": "pswin_win_attn_fused_fwd (inference)", "win_fused_fwd_kernel<true>": "pswin_win_attn_fused_fwd (training saves)",
         "qkv_attn_fwd_kernel<192, false>": "pswin_qkv_attn_fused_fwd C=192 (inference)", "qkv_attn_fwd_kernel<192, true>": "pswin_qkv_attn_fused_fwd C=192 (training saves)",
         "qkv_attn_fwd_kernel<384, false>": "pswin_qkv_attn_fused_fwd C=384 (inference)", "qkv_attn_fwd_kernel<384, true>": "pswin_qkv_attn_fused_fwd C=384 (training saves)",
         "attn_fwd_kernel": "pswin_attn_fwd", "attn_bwd_pair_kernel": "pswin_attn_bwd"}
# in-kernel clock under the kernel's own load, GHz (profiles/r03_fused_window_clock_probe.txt)
MEASURED_GHZ = {"pswin_win_attn_fused_fwd (inference)": 2.365, "pswin_win_attn_fused_fwd (training saves)": 2.257}
out = {"lib_digest": bench._lib_digest(), "counter": "SQ_VALU_MFMA_BUSY_CYCLES (rocprofv3 --pmc, own pass)",
       "measured_clock_source": "profiles/r03_fused_window_clock_probe.txt (median over workgroups of d s_memtime / d s_memrealtime x 100 MHz)", "kernels": {}}


def label(kernel_name):
    """the NAMES entry with the longest key contained in the kernel name (qkv_attn_fwd_kernel contains attn_fwd_kernel)"""
    hits = [k for k in NAMES if k in kernel_name]
    return NAMES[max(hits, key=len)] if hits else None


args = sys.argv[1:]
outp = args[2]
pairs = [(args[0], args[1])] + [(args[i], args[i + 1]) for i in range(3, len(args) - 1, 2)]
for pmc, trace in pairs:
    busy, dur = defaultdict(list), defaultdict(list)
    for r in csv.DictReader(open(pmc)):
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES" and label(r["Kernel_Name"]):
            busy[label(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for r in csv.DictReader(open(trace)):
        name = r.get("Kernel_Name") or r.get("Name")
        n = label(name)
        if n:
            dur[n].append(float(r["AverageNs"]) if "AverageNs" in r else float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for n in busy:
        if not dur[n]:
            continue
        b, d = sum(busy[n]) / len(busy[n]), sum(dur[n]) / len(dur[n])
        out["kernels"][n] = {"mfma_busy_cycles_per_launch": b, "avg_launch_ns": d, "mfma_busy_at_2.4GHz": round(b / (d * 1e-9 * 2.4e9 * 1024), 4),
                             "mfma_busy_at_2.0GHz": round(b / (d * 1e-9 * 2.0e9 * 1024), 4)}
        if n in MEASURED_GHZ:
            out["kernels"][n]["measured_clock_GHz"] = MEASURED_GHZ[n]
            out["kernels"][n]["mfma_busy_at_measured_clock"] = round(b / (d * 1e-9 * MEASURED_GHZ[n] * 1e9 * 1024), 4)
json.dump(out, open(outp, "w"), indent=1)
print(json.dumps(out, indent=1))
