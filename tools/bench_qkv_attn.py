"""Micro-benchmark of the fused qkv + attention-core kernel (csrc/pswin_qkvattn.hip) against the two kernels it replaces (qkv GEMM ->
pswin_attn_fwd), at the PanoSwin-T stage-1 / stage-2 shapes (C = 192 / 384), bf16, each replayed 30x from a hipGraph.
usage: python tools/bench_qkv_attn.py [B]"""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402,F401  (shipped hipBLASLt table for the library legs)
from panoswintransformerobjectdetection_amd import ops  # noqa: E402
from panoswintransformerobjectdetection_amd.backbone import WindowAttention  # noqa: E402
sys.path.insert(0, "tools")
from bench_gemm_nt import t  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda:0"
for (H, W, C) in ((64, 128, 192), (32, 64, 384)):
    heads = C // 32
    torch.manual_seed(0)
    att = WindowAttention(C, 7, heads).to(dev)
    for lin in (att.qkv, att.proj):
        lin.__dict__["_lowp"] = (lin.weight.detach().to(torch.bfloat16), lin.bias.detach().to(torch.bfloat16))
    for shift in (0, 3):
        tiles = ops.window_dist_tiles(H, W, shift, dev)
        nW = tiles.n
        n = B * nW
        x = torch.randn(n * 49, C, device=dev).to(torch.bfloat16)
        flop = n * (2 * 49 * C * 3 * C + heads * 4 * 49 * 49 * 32)            # algorithmic, 49 tokens
        issued = n * (2 * 64 * C * 3 * C + heads * 4 * 64 * 64 * 32)
        with torch.no_grad():
            t_inf = t(lambda: ops.window_attention_qkv_fused(x, att, tiles, None, nW))
        xg = x.clone().requires_grad_(True)
        t_train = t(lambda: ops.window_attention_qkv_fused(xg, att, tiles, None, nW))

        def chain():
            qkv = ops.linear(x, att.qkv, torch.bfloat16)
            return ops.window_attention(qkv, att.sphere_position_alpha_table_Te, att.sphere_position_beta_table_Te, tiles, None, heads, att.scale, nW)
        with torch.no_grad():
            t_chain = t(chain)
        print(f"C {C} tokens {H}x{W} shift {shift} windows {n}: fused inference {t_inf:6.1f} us ({flop / t_inf * 1e-6:6.1f} TFLOP/s algorithmic = "
              f"{flop / t_inf * 1e-6 / 25:4.1f} % of 2.5 PF; issued {issued / t_inf * 1e-6:6.1f}) | fused training {t_train:6.1f} us "
              f"({flop / t_train * 1e-6 / 25:4.1f} %) | qkv GEMM + pswin_attn_fwd {t_chain:6.1f} us", flush=True)
