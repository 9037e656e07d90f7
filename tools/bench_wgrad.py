"""Weight-gradient GEMM dW = dY^T X (contraction over M rows): chunked bmm + column sum (current) vs one GEMM with
TunableOp picking the hipBLASLt / rocBLAS solution (incl. stream-K)."""
import os, sys, torch
os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"; os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
os.environ["PYTORCH_TUNABLEOP_FILENAME"] = "gpurun_out/tune_wgrad/results.csv"
os.makedirs("gpurun_out/tune_wgrad", exist_ok=True)
sys.path.insert(0, ".")
from panoswintransformerobjectdetection_amd import ops
from panoswintransformerobjectdetection_amd.backbone import _pick_split
dev = "cuda:0"
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (M, N, K) in [(275576, 288, 96), (275576, 96, 96), (262144, 384, 96), (262144, 96, 384), (74480, 576, 192), (65536, 768, 192), (19600, 1152, 384), (16384, 1536, 384), (5880, 2304, 768), (4096, 3072, 768)]:
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16); x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    ch = _pick_split(M, -(-N // 64) * -(-K // 64))
    def cur():
        if ch > 1:
            part = torch.bmm(dy.view(ch, M // ch, N).transpose(1, 2), x.view(ch, M // ch, K))
            return ops.colsum(part.view(ch, N * K)).view(N, K)
        return (dy.t() @ x).float()
    def single():
        return (dy.t() @ x).float()
    out32 = torch.empty(N, K, device=dev)
    print(f"M={M} N={N} K={K} ch={ch}: bmm+colsum {t(cur):.1f} us | single GEMM (tuned) {t(single):.1f} us | ideal {(M*(N+K)*2)/4.7e6:.1f} us", flush=True)
