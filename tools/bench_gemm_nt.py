"""The tiled HIP GEMM (pswin_gemm_nt) against the library GEMM PyTorch-ROCm selects (shipped TunableOp table), forward and
data-gradient shapes of PanoSwin-T stages 1-3 at batch 8, each replayed 30x from a hipGraph.  usage: python tools/bench_gemm_nt.py"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402,F401  (installs the shipped hipBLASLt solution table)

import torch
import torch.nn.functional as F

from panoswintransformerobjectdetection_amd import ops

dev = "cuda:0"


def t(fn, n=30):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


shapes = [("m1 red", 65536, 384, 192, 1), ("s1 qkv", 74480, 192, 576, 2), ("s1 proj", 74480, 192, 192, 2), ("s1 fc1", 65536, 192, 768, 2),
          ("s1 fc2", 65536, 768, 192, 2), ("m2 red", 16384, 768, 384, 1), ("s2 qkv", 19600, 384, 1152, 6), ("s2 proj", 19600, 384, 384, 6),
          ("s2 fc1", 16384, 384, 1536, 6), ("s2 fc2", 16384, 1536, 384, 6), ("m3 red", 4096, 1536, 768, 1), ("s3 qkv", 5880, 768, 2304, 2),
          ("s3 proj", 5880, 768, 768, 2), ("s3 fc1", 4096, 768, 3072, 2), ("s3 fc2", 4096, 3072, 768, 2)]
if len(sys.argv) > 1:                     # python tools/bench_gemm_nt.py <batch>: the same shapes at another batch size (rows scale with it)
    _b = int(sys.argv[1])
    shapes = [(n, M * _b // 8, K, N, c) for n, M, K, N, c in shapes]


def main():
    tot = {"lib_fwd": 0.0, "hip_fwd": 0.0, "ring_fwd": 0.0, "lib_dgrad": 0.0, "hip_dgrad": 0.0, "ring_dgrad": 0.0, "floor": 0.0}
    print(f"{'':8s} {'M':>6s} {'K':>5s} {'N':>5s} | fwd: lib  hip64 hip128   ring | dgrad: lib  hip64 hip128   ring | floor us | x count")
    for name, M, K, N, cnt in shapes:
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        wt = w.t().contiguous()
        dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
        lf = t(lambda: F.linear(x, w))
        ld = t(lambda: dy @ w)
        hf = [t(lambda: ops.gemm_nt(x, w, None, tm)) if ops.gemm_nt_supported(x, N) else float("nan") for tm in (64, 128)]
        hd = [t(lambda: ops.gemm_nt(dy, wt, None, tm)) if ops.gemm_nt_supported(dy, K) else float("nan") for tm in (64, 128)]
        fl = max(2 * (M * K + M * N + N * K) / 6.3e12, 2.0 * M * K * N / 2.5e15) * 1e6
        rf = rd = float("nan")            # (the persistent ring variant of round 3 was removed in round 4: profiles/r03_gemm_nt_ring_vs_tiled.txt)
        print(f"{name:8s} {M:6d} {K:5d} {N:5d} | {lf:8.1f} {hf[0]:6.1f} {hf[1]:6.1f} {rf:6.1f} | {ld:10.1f} {hd[0]:6.1f} {hd[1]:6.1f} {rd:6.1f} | {fl:6.1f}   | x{cnt}", flush=True)
        tot["ring_fwd"] += (rf if rf == rf else lf) * cnt
        tot["ring_dgrad"] += (rd if rd == rd else ld) * cnt
        best = lambda v: min(z for z in v if z == z) if any(z == z for z in v) else float("nan")
        tot["lib_fwd"] += lf * cnt
        tot["lib_dgrad"] += ld * cnt
        tot["hip_fwd"] += (best(hf) if best(hf) == best(hf) else lf) * cnt
        tot["hip_dgrad"] += (best(hd) if best(hd) == best(hd) else ld) * cnt
        tot["floor"] += fl * cnt
    print("per step (us):", {k: round(v) for k, v in tot.items()})



if __name__ == "__main__":
    main()
