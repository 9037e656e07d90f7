"""Launch the fused window kernel at the PanoSwin-T stage-0 shape (B = 8, bf16) a few times: target for rocprofv3 --pmc /
--kernel-trace.  usage: python tools/pmc_fused.py [train|infer] [launches]"""
import sys

import torch

sys.path.insert(0, ".")
from panoswintransformerobjectdetection_amd import ops  # noqa: E402
from panoswintransformerobjectdetection_amd.backbone import WindowAttention  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "infer"
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = "cuda:0"
H, W, heads, C, B = 128, 256, 3, 96, 8
torch.manual_seed(0)
att = WindowAttention(C, 7, heads).to(dev)
for lin in (att.qkv, att.proj):
    lin.__dict__["_lowp"] = (lin.weight.detach().to(torch.bfloat16), lin.bias.detach().to(torch.bfloat16))
tiles = ops.window_dist_tiles(H, W, 3, dev)
nW = tiles.n
x = torch.randn(B * nW * 49, C, device=dev).to(torch.bfloat16)
if mode == "train":
    x.requires_grad_(True)
for _ in range(launches):
    if mode == "train":
        ops.window_attention_fused(x, att, tiles, None, nW)
    else:
        with torch.no_grad():
            ops.window_attention_fused(x, att, tiles, None, nW)
torch.cuda.synchronize()
n = B * nW
print("algorithmic FLOP per launch", n * (2 * 49 * C * 3 * C + heads * 4 * 49 * 49 * 32 + 2 * 49 * C * C), "windows", n)
