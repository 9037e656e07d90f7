"""Launch the fused qkv + attention kernel at the PanoSwin-T stage-1 (C = 192) or stage-2 (C = 384) shape (B = 8, bf16), inference and
training variant, a few times each: target for rocprofv3 --pmc / --kernel-trace.  usage: python tools/pmc_qkv_attn.py [192|384] [launches]"""
import sys

import torch

sys.path.insert(0, ".")
from panoswintransformerobjectdetection_amd import ops  # noqa: E402
from panoswintransformerobjectdetection_amd.backbone import WindowAttention  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 192
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = "cuda:0"
H, W = (64, 128) if C == 192 else (32, 64)
heads, B = C // 32, 8
torch.manual_seed(0)
att = WindowAttention(C, 7, heads).to(dev)
for lin in (att.qkv, att.proj):
    lin.__dict__["_lowp"] = (lin.weight.detach().to(torch.bfloat16), lin.bias.detach().to(torch.bfloat16))
tiles = ops.window_dist_tiles(H, W, 3, dev)
nW = tiles.n
x = torch.randn(B * nW * 49, C, device=dev).to(torch.bfloat16)
xg = x.clone().requires_grad_(True)
for _ in range(launches):
    with torch.no_grad():
        ops.window_attention_qkv_fused(x, att, tiles, None, nW)
    ops.window_attention_qkv_fused(xg, att, tiles, None, nW)
torch.cuda.synchronize()
n = B * nW
print("algorithmic FLOP per launch", n * (2 * 49 * C * 3 * C + heads * 4 * 49 * 49 * 32), "windows", n)
