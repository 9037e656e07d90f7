"""The HIP weight-gradient kernel (pswin_gemm_tn) against the library path (torch.bmm over row chunks, the split-K form the
backbone used for every weight gradient), PanoSwin-T stage 1-3 shapes at batch 8, each replayed 30x from a hipGraph.
usage: python tools/bench_gemm_tn.py"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402,F401  (installs the shipped hipBLASLt solution table)

import torch

from panoswintransformerobjectdetection_amd import _lib, ops
from panoswintransformerobjectdetection_amd.ops import _pick_split
from bench_gemm_nt import shapes, t

dev = "cuda:0"
lib = _lib.load()
tot = {"lib": 0.0, "hip": 0.0, "ring_f32": 0.0, "ring_bf16": 0.0, "floor": 0.0}
print(f"{'':8s} {'M':>6s} {'K':>5s} {'N':>5s} | lib bmm (ch) | hip tn (splits) | ring f32 / bf16 (splits) | floor us | x count")
shapes = [("s0 qkv", 275576, 96, 288, 2), ("s0 proj", 275576, 96, 96, 2), ("s0 fc1", 262144, 96, 384, 2), ("s0 fc2", 262144, 384, 96, 2)] + list(shapes)
for name, M, K, N, cnt in shapes:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
    ch = _pick_split(M, -(-N // 64) * -(-K // 64))
    lt = t(lambda: torch.bmm(dy.view(ch, M // ch, N).transpose(1, 2), x.view(ch, M // ch, K))) if ch > 1 else t(lambda: dy.t() @ x)
    ht, sp = float("nan"), 0
    if lib.pswin_gemm_tn_supported(M, N, K):
        sp = lib.pswin_gemm_tn_splits(M, N, K)
        ht = t(lambda: ops.gemm_tn(dy, x, sp))
        if "--sweep" in sys.argv:
            tiles = (N // 64) * (K // 64) // 6
            res = []
            for wgs in (128, 256, 384, 512, 768, 1024):
                s2 = max(1, min(M // 64, -(-wgs // tiles)))
                res.append((s2, round(t(lambda: ops.gemm_tn(dy, x, s2)), 1)))
            print("      sweep (splits, us):", res, flush=True)
    fl = max((2 * (M * K + M * N) + 4 * N * K) / 6.3e12, 2.0 * M * K * N / 2.5e15) * 1e6
    rf = rb = float("nan")
    rs = 0
    if lib.pswin_gemm_tn_ring_supported(M, N, K):
        rs = lib.pswin_gemm_tn_ring_splits(M, N, K, 0)
        rf = t(lambda: ops.gemm_tn_ring(dy, x, rs, torch.float32))
        rb = t(lambda: ops.gemm_tn_ring(dy, x, rs, torch.bfloat16))
        if "--sweep" in sys.argv:
            tiles = (N // 192) * (K // 192) if K % 192 == 0 and N % 192 == 0 else 1
            res = []
            for wgs in (128, 192, 256, 384, 512):
                s2 = max(1, min(M // 64, wgs // tiles))
                res.append((s2, round(t(lambda: ops.gemm_tn_ring(dy, x, s2, torch.bfloat16)), 1)))
            print("      ring sweep (splits, us, bf16 slabs):", res, flush=True)
    print(f"{name:8s} {M:6d} {K:5d} {N:5d} | {lt:7.1f} ({ch:3d}) | {ht:7.1f} ({sp:3d})   | {rf:7.1f} / {rb:7.1f} ({rs:3d}) | {fl:6.1f}   | x{cnt}", flush=True)
    tot["lib"] += lt * cnt
    tot["ring_f32"] += (rf if rf == rf else lt) * cnt
    tot["ring_bf16"] += (rb if rb == rb else lt) * cnt
    tot["hip"] += (ht if ht == ht else lt) * cnt
    tot["floor"] += fl * cnt
print("per step (us):", {k: round(v) for k, v in tot.items()})
