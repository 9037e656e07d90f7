"""Scratch micro-benchmarks (GPU box): weight-gradient GEMM strategies and PatchEmbed precision/layout."""
import time
import torch
import torch.nn as nn
import torch.nn.functional as F

dev = "cuda:0"


def bench(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


print("== wgrad: dW[N,K] = dY[M,N]^T X[M,K], bf16 ==")
for (M, N, K) in [(275576, 288, 96), (275576, 96, 96), (262144, 384, 96), (262144, 96, 384), (74480, 576, 192),
                  (65536, 768, 192), (19600, 1152, 384), (16384, 1536, 384), (5880, 2304, 768), (4096, 3072, 768)]:
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    t_mm = bench(lambda: dy.t() @ x)
    res = {}
    for ch in (8, 16, 32, 64, 128):
        if M % ch:
            continue
        def f():
            p = torch.bmm(dy.view(ch, M // ch, N).transpose(1, 2), x.view(ch, M // ch, K))
            return p.sum(0)
        res[ch] = bench(f)
    def f32acc():
        p = torch.bmm(dy.view(8, M // 8, N).transpose(1, 2), x.view(8, M // 8, K))
        return p.float().sum(0)
    ideal = (M * (N + K) * 2) / 5e12 * 1e6
    print(f"M={M} N={N} K={K}: mm {t_mm:.0f}us | bmm-splitK " + " ".join(f"{c}:{t:.0f}" for c, t in res.items()) + f" | HBM-ideal {ideal:.0f}us")
    # bias grad
    t_b = bench(lambda: dy.sum(0))
    t_b32 = bench(lambda: dy.float().sum(0))
    print(f"    bias-grad sum(0): {t_b:.0f}us  (via float {t_b32:.0f}us)")

print("== fwd / dgrad GEMMs ==")
for (M, N, K) in [(275576, 288, 96), (262144, 384, 96), (262144, 96, 384), (16384, 1536, 384)]:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    t_f = bench(lambda: F.linear(x, w, b))
    t_d = bench(lambda: dy @ w)
    print(f"M={M} N={N} K={K}: fwd {t_f:.0f}us ({2*M*N*K/t_f/1e6:.0f} TF/s)  dgrad {t_d:.0f}us | HBM-ideal fwd {(M*(N+K)*2)/5e12*1e6:.0f}us")

print("== PatchEmbed fwd+bwd, B=8 3x512x1024 ==")
class PE(nn.Module):
    def __init__(s):
        super().__init__()
        s.proj = nn.Sequential(nn.Conv2d(3, 32, 3, 1, 1), nn.BatchNorm2d(32), nn.ReLU(inplace=True),
                               nn.Conv2d(32, 64, 3, 1, 1), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
                               nn.Conv2d(64, 96, 4, 4))
    def forward(s, x):
        return s.proj(x)
pe = PE().to(dev).train()
x = torch.randn(8, 3, 512, 1024, device=dev)
for name, cl, ac in [("fp32 NCHW", False, False), ("fp32 NHWC", True, False), ("bf16 autocast NCHW", False, True), ("bf16 autocast NHWC", True, True)]:
    xin = x.contiguous(memory_format=torch.channels_last) if cl else x
    m = pe.to(memory_format=torch.channels_last) if cl else pe.to(memory_format=torch.contiguous_format)
    def f():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=ac):
            y = m(xin)
        y.float().mean().backward()
    try:
        print(f"{name}: {bench(f, n=5, warm=2)/1e3:.2f} ms")
    except Exception as ex:
        print(name, "failed:", repr(ex)[:200])
