"""bias+GELU kernels against torch's GELU kernels, C calls only (no allocation in the timed loop)."""
import sys, torch
sys.path.insert(0, ".")
import torch.nn.functional as F
from panoswintransformerobjectdetection_amd import _lib
dev = "cuda:0"
lib = _lib.load()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
flush = torch.empty(1 << 28, device=dev)
for M, N in [(262144, 384), (65536, 768), (16384, 1536), (4096, 3072)]:
    y = torch.randn(M, N, device=dev).to(torch.bfloat16); b = torch.randn(N, device=dev); g = torch.randn(M, N, device=dev).to(torch.bfloat16)
    out = torch.empty_like(y); db = torch.empty(N, device=dev)
    ws = torch.empty(lib.pswin_bias_gelu_workspace(M, N), device=dev)
    yb = y.clone().requires_grad_(True)
    tf = t(lambda: F.gelu(yb))
    o = F.gelu(yb)
    tb = t(lambda: torch.autograd.grad(o, yb, g, retain_graph=True))
    line = f"M={M} N={N}: torch fwd {tf:.1f} bwd {tb:.1f} |"
    for uf, ub in [(1, 1), (2, 2), (4, 4)]:
        assert lib.pswin_bias_gelu_tune(uf, ub) == 0
        mf = t(lambda: _lib.call("pswin_bias_gelu_fwd", y, y.data_ptr(), 1, b.data_ptr(), out.data_ptr(), M, N))
        mb = t(lambda: _lib.call("pswin_bias_gelu_bwd", y, g.data_ptr(), y.data_ptr(), 1, b.data_ptr(), out.data_ptr(), db.data_ptr(), ws.data_ptr(), M, N))
        line += f" unr{uf}: fwd {mf:.1f} bwd {mb:.1f} |"
    print(line)
