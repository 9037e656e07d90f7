"""Weight-gradient GEMM dW = dY^T X (contraction over M rows) as a chunked bmm + column sum: sweep of the chunk count."""
import os, sys, torch
os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1"); os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "1")
os.makedirs("gpurun_out/tune_wgrad", exist_ok=True)
os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", "gpurun_out/tune_wgrad/results.csv")
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
for (M, N, K) in [(275576, 288, 96), (262144, 384, 96), (262144, 96, 384), (74480, 576, 192), (65536, 768, 192), (65536, 192, 768),
                  (19600, 1152, 384), (16384, 1536, 384), (16384, 384, 1536), (5880, 2304, 768), (4096, 3072, 768), (4096, 768, 3072)]:
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16); x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    cur = _pick_split(M, -(-N // 64) * -(-K // 64))
    res = []
    for ch in sorted({c for c in range(1, 513) if M % c == 0 and (c in (1, 2, 4, 8, 16, 32, 64, 128, 256) or c == cur or abs(c - cur) <= max(2, cur // 3))}):
        if M // ch < 256: continue
        def run():
            if ch > 1:
                part = torch.bmm(dy.view(ch, M // ch, N).transpose(1, 2), x.view(ch, M // ch, K))
                return ops.colsum(part.view(ch, N * K)).view(N, K)
            return (dy.t() @ x).float()
        res.append((t(run), ch))
    res.sort()
    print(f"M={M} N={N} K={K} current ch={cur}: " + " ".join(f"ch{c}:{us:.0f}" for us, c in res[:6]) + f" | cur: {[f'{us:.0f}' for us, c in res if c == cur]}", flush=True)
