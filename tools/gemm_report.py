"""Library GEMMs of one PanoSwin-T step (batch 8, 512x1024) timed one by one as the backbone issues them, against the
per-GEMM roofline max(bytes / 6.3 TB/s, flops / 2.5 PFLOP/s).  usage: python tools/gemm_report.py"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402,F401  (importing it installs the shipped hipBLASLt solution table: bench.use_shipped_gemm_table)

import torch
import torch.nn.functional as F

sys.path.insert(0, ROOT)
from panoswintransformerobjectdetection_amd.backbone import _pick_split

dev = "cuda:0"


def t(fn, n=30):
    """n back-to-back launches inside one hipGraph (no host launch cost, as in the captured training step)"""
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
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


# (name, M rows, K in, N out, count per step)
shapes = [("s0 qkv", 275576, 96, 288, 2), ("s0 proj", 275576, 96, 96, 2), ("s0 fc1", 262144, 96, 384, 2), ("s0 fc2", 262144, 384, 96, 2),
          ("m1 red", 65536, 384, 192, 1),
          ("s1 qkv", 74480, 192, 576, 2), ("s1 proj", 74480, 192, 192, 2), ("s1 fc1", 65536, 192, 768, 2), ("s1 fc2", 65536, 768, 192, 2),
          ("m2 red", 16384, 768, 384, 1),
          ("s2 qkv", 19600, 384, 1152, 6), ("s2 proj", 19600, 384, 384, 6), ("s2 fc1", 16384, 384, 1536, 6), ("s2 fc2", 16384, 1536, 384, 6),
          ("m3 red", 4096, 1536, 768, 1),
          ("s3 qkv", 5880, 768, 2304, 2), ("s3 proj", 5880, 768, 768, 2), ("s3 fc1", 4096, 768, 3072, 2), ("s3 fc2", 4096, 3072, 768, 2)]
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
floor = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
print(f"{'':8s} {'M':>7s} {'K':>5s} {'N':>5s} | {'fwd':>6s} {'dgrad':>6s} {'wgrad':>6s} (ch) | floor us | x count")
for name, M, K, N, cnt in shapes:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
    ch = _pick_split(M, -(-N // 64) * -(-K // 64))
    f = t(lambda: F.linear(x, w))
    d = t(lambda: dy @ w)
    if ch > 1:
        g = t(lambda: torch.bmm(dy.view(ch, M // ch, N).transpose(1, 2), x.view(ch, M // ch, K)))
    else:
        g = t(lambda: dy.t() @ x)
    fl = max(2 * (M * K + M * N + N * K) / 6.3e12, 2.0 * M * K * N / 2.5e15) * 1e6
    print(f"{name:8s} {M:7d} {K:5d} {N:5d} | {f:6.1f} {d:6.1f} {g:6.1f} ({ch:3d}) | {fl:6.1f}   | x{cnt}", flush=True)
    for k, v in (("fwd", f), ("dgrad", d), ("wgrad", g)):
        tot[k] += v * cnt
        floor[k] += fl * cnt
print("per step (us):", {k: round(v) for k, v in tot.items()}, "floors:", {k: round(v) for k, v in floor.items()})
