import sys, torch
sys.path.insert(0, ".")
import torch.nn.functional as F
from panoswintransformerobjectdetection_amd import ops
dev = "cuda:0"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for name, M, K, N in [("s0 qkv", 275576, 96, 288), ("s0 proj", 275576, 96, 96), ("s0 fc1", 262144, 96, 384), ("s0 fc2", 262144, 384, 96), ("s1 proj", 74480, 192, 192)]:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.1).to(torch.bfloat16); dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev); bb = bias.to(torch.bfloat16)
    a = t(lambda: F.linear(x, w, bb)); b = t(lambda: ops.skinny_gemm(x, w, bias))
    line = f"{name}: fwd lib {a:.1f} us, skinny {b:.1f} us (ideal {2*M*(K+N)/4.7e6:.0f})"
    if ops._lib.load().pswin_gemm_skinny_supported(N, K):
        c = t(lambda: dy @ w); d = t(lambda: ops.skinny_gemm(dy, w, None, True))
        line += f" | dgrad lib {c:.1f} us, skinny {d:.1f} us"
    print(line, flush=True)
# fused fc1 + GELU (stage 0)
from panoswintransformerobjectdetection_amd import _lib
M, K, N = 262144, 96, 384
x = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.1).to(torch.bfloat16); b = torch.randn(N, device=dev)
h = torch.empty(M, N, device=dev, dtype=torch.bfloat16); dh = torch.randn(M, N, device=dev).to(torch.bfloat16); dy = torch.empty_like(h)
db = torch.empty(N, device=dev); ws = torch.empty(_lib.load().pswin_fc1_gelu_workspace(N), device=dev)
f = t(lambda: _lib.call("pswin_fc1_gelu_fwd", x, x.data_ptr(), w.data_ptr(), b.data_ptr(), h.data_ptr(), M, K, N))
g = t(lambda: _lib.call("pswin_fc1_gelu_bwd", x, x.data_ptr(), w.data_ptr(), b.data_ptr(), dh.data_ptr(), dy.data_ptr(), db.data_ptr(), ws.data_ptr(), M, K, N))
print(f"fc1+gelu fused: fwd {f:.1f} us, bwd {g:.1f} us")
