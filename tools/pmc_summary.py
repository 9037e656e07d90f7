"""Combine two rocprofv3 --pmc runs of bench.py (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM traffic per launch.

usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

Corrections (MI355X_MICROARCH.md, HBM section): counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of
the bytes of a wide (16 B/lane) coalesced streaming read (128-byte requests tallied at 64 B) -> fetch x 2 for the
kernels that read whole rows; "other access widths are uncalibrated", so the attention kernels' pattern (64-byte
head segments of rows 576+ B apart) was calibrated on known byte counts with tools/pmc_calib.py: a 64-byte segment
per row reads back EXACT (17.7 MB counted for 17.6 MB), the wide clone of the same tensor half (79.4 for 158.7 MB)
-> factor 1 for pswin_attn_fwd / pswin_attn_bwd.  WRITE_SIZE is exact for 16 B/lane streaming stores.  The first launches (warm-up, lazily built tables) are included; they are
the same kernels on the same shapes.
"""
import csv, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

KERNELS = {"win_fused_fwd_kernel": "pswin_win_attn_fused_fwd", "attn_fwd_kernel": "pswin_attn_fwd", "attn_bwd_kernel": "pswin_attn_bwd_ex", "attn_bwd_pair_kernel": "pswin_attn_bwd_ex", "ln_fwd_kernel": "pswin_ln_gather_fwd",
           "ln_bwd_kernel": "pswin_ln_gather_bwd", "window_gather_kernel": "pswin_window_gather", "window_gather8_kernel": "pswin_window_gather",
           "window_scatter_add_kernel": "pswin_window_scatter_add", "window_scatter_add8_kernel": "pswin_window_scatter_add",
           "ln_add_fwd_kernel": "pswin_scatter_add_ln_fwd",
           # the tiled GEMM by epilogue (template arguments <rows, EPI>): plain, fused GELU backward, fused GELU forward
           # (round 4: a third template argument = LDS stages; 96-row tiles; EPI 3 = f32 result)
           "gemm_nt_kernel<64, 0,": "pswin_gemm_nt", "gemm_nt_kernel<96, 0,": "pswin_gemm_nt", "gemm_nt_kernel<128, 0,": "pswin_gemm_nt",
           "gemm_nt_kernel<64, 3,": "pswin_gemm_nt", "gemm_nt_kernel<128, 3,": "pswin_gemm_nt",
           "gemm_nt_kernel<64, 1,": "pswin_gemm_nt_gelu_bwd", "gemm_nt_kernel<128, 1,": "pswin_gemm_nt_gelu_bwd",
           "gemm_nt_kernel<64, 2,": "pswin_gemm_nt_gelu_fwd", "gemm_nt_kernel<128, 2,": "pswin_gemm_nt_gelu_fwd",
           # round 3
           "gemm_tn_ring_kernel": "pswin_gemm_tn_ring", "qkv_attn_fwd_kernel": "pswin_qkv_attn_fused_fwd", "mlp0_fwd_kernel": "pswin_mlp0_fwd",
           "mlp0_bwd_kernel": "pswin_mlp0_bwd", "skinny_gemm_kernel": "pswin_gemm_skinny"}


# entry points whose ONE call launches one kernel per template instantiation seen (round 4: the grouped weight-gradient call = one
# gemm_tn_ring_kernel launch per tile geometry): their "launch" below is the call, i.e. kernel launches / distinct instantiations
PER_CALL = {"pswin_gemm_tn_ring"}


def collect(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    variants = defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        hits = [key for key in KERNELS if key in r["Kernel_Name"]]
        if hits:                                     # the longest key wins (qkv_attn_fwd_kernel contains attn_fwd_kernel)
            name = KERNELS[max(hits, key=len)]
            acc[name][0] += float(r["Counter_Value"]) * 1024.0
            acc[name][1] += 1
            variants[name].add(r["Kernel_Name"])
    for name in PER_CALL:
        if name in acc and len(variants[name]) > 1:
            acc[name][1] = max(1, acc[name][1] // len(variants[name]))
    return acc


FETCH_FACTOR = {"pswin_attn_fwd": 1.0, "pswin_attn_bwd_ex": 1.0, "pswin_win_attn_fused_fwd": 2.0,      # calibrated (see above); default 2.0 (wide row reads)
                "pswin_qkv_attn_fused_fwd": 1.0}                                                            # 64-byte segments of a row per instruction, as the attention kernels
fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {}
for name in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(name, [0, 1])
    w, nw = write.get(name, [0, 1])
    k = FETCH_FACTOR.get(name, 2.0)
    out[name] = {"launches": nf, "fetch_bytes_per_launch_raw": f / max(nf, 1), "fetch_correction": k,
                 "fetch_bytes_per_launch_corrected": k * f / max(nf, 1),
                 "write_bytes_per_launch": w / max(nw, 1), "hbm_bytes_per_launch": k * f / max(nf, 1) + w / max(nw, 1)}
import bench  # noqa: E402  (kernel-source digest: bench.py quotes these numbers only for the same sources)
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on bench.py", "lib_digest": bench._lib_digest(),
           "kernels": out}, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
