"""Forward / data-gradient GEMMs of the Linear layers: weight stored [N, K] (nn.Linear layout) vs pre-transposed [K, N];
TunableOp picks the best solution for each."""
import os, sys, torch
os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"; os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
os.makedirs("gpurun_out/tune_layout", exist_ok=True)
os.environ["PYTORCH_TUNABLEOP_FILENAME"] = "gpurun_out/tune_layout/results.csv"
import torch.nn.functional as F
dev = "cuda:0"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
shapes = [("s0 qkv", 275576, 96, 288), ("s0 proj", 275576, 96, 96), ("s0 fc1", 262144, 96, 384), ("s0 fc2", 262144, 384, 96),
          ("s1 qkv", 74480, 192, 576), ("s1 fc1", 65536, 192, 768), ("s1 fc2", 65536, 768, 192),
          ("s2 qkv", 19600, 384, 1152), ("s2 fc1", 16384, 384, 1536), ("s2 fc2", 16384, 1536, 384),
          ("s3 fc1", 4096, 768, 3072), ("s3 fc2", 4096, 3072, 768)]
for name, M, K, N in shapes:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16); w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    wt = w.t().contiguous(); dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
    f_nt = t(lambda: F.linear(x, w)); f_nn = t(lambda: x @ wt)
    d_nn = t(lambda: dy @ w); d_nt = t(lambda: F.linear(dy, wt))
    print(f"{name:8s} M={M} K={K} N={N}: fwd [N,K] {f_nt:6.1f} | fwd [K,N] {f_nn:6.1f} || dgrad [N,K] {d_nn:6.1f} | dgrad [K,N] {d_nt:6.1f} us", flush=True)
