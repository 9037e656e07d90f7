"""Launch pswin_gemm_nt on one shape a few times: target for rocprofv3 --pmc.  usage: python tools/pmc_gemm_nt.py M K N tile_m"""
import sys

import torch

sys.path.insert(0, ".")
from panoswintransformerobjectdetection_amd import ops  # noqa: E402

M, K, N, tm = [int(v) for v in sys.argv[1:5]]
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
for _ in range(6):
    ops.gemm_nt(x, w, None, tm)
torch.cuda.synchronize()
print("flop", 2 * M * K * N)
