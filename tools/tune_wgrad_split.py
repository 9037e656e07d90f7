"""Pick the split count of the chunked weight-gradient GEMM (backbone._LinearSplitK.backward) for the PanoSwin-T shapes at
B = 8, 512x1024: TunableOp tunes every candidate batched GEMM, 3 timing rounds of 15 iterations each, median.
Prints a python dict for backbone._SPLIT_TABLE; the TunableOp results go to gpurun_out/tune_split/."""
import os, sys, statistics, torch
os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"; os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
os.makedirs("gpurun_out/tune_split", exist_ok=True)
os.environ["PYTORCH_TUNABLEOP_FILENAME"] = "gpurun_out/tune_split/results.csv"
sys.path.insert(0, ".")
from panoswintransformerobjectdetection_amd import ops
dev = "cuda:0"
def t(fn, n=15):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
shapes = []
for C, Mw, Mt in [(96, 275576, 262144), (192, 74480, 65536), (384, 19600, 16384), (768, 5880, 4096)]:
    shapes += [(Mw, 3 * C, C), (Mw, C, C), (Mt, 4 * C, C), (Mt, C, 4 * C)]
shapes += [(65536, 192, 384), (16384, 384, 768), (4096, 768, 1536)]
table = {}
for (M, N, K) in shapes:
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16); x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    cands = [c for c in range(1, 513) if M % c == 0 and M // c >= 384]
    keep = sorted({min(cands, key=lambda c: abs(c - w)) for w in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384)})
    res = []
    for ch in keep:
        def run():
            if ch > 1:
                part = torch.bmm(dy.view(ch, M // ch, N).transpose(1, 2), x.view(ch, M // ch, K))
                return ops.colsum(part.view(ch, N * K)).view(N, K)
            return (dy.t() @ x).float()
        res.append((statistics.median([t(run) for _ in range(3)]), ch))
    res.sort()
    table[(M, N, K)] = res[0][1]
    print(f"# M={M} N={N} K={K}: " + " ".join(f"ch{c}:{us:.0f}" for us, c in res[:5]), flush=True)
print("_SPLIT_TABLE = {" + ", ".join(f"({m}, {n}, {k}): {c}" for (m, n, k), c in table.items()) + "}")
