"""Time the fused stem kernels at the PanoSwin-T bench shape (B = 8, 512x1024)."""
import sys, torch
sys.path.insert(0, ".")
from panoswintransformerobjectdetection_amd import stem
dev = "cuda:0"
B, H, W = 8, 512, 1024
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 3, H, W, generator=g).to(dev)
w1 = (torch.randn(32, 3, 3, 3, generator=g) * 0.3).to(dev); w2 = (torch.randn(64, 32, 3, 3, generator=g) * 0.08).to(dev)
w3 = (torch.randn(96, 64, 4, 4, generator=g) * 0.04).to(dev); b3 = torch.zeros(96, device=dev)
sc1 = torch.ones(32, device=dev); sh1 = torch.zeros(32, device=dev); sc2 = torch.ones(64, device=dev); sh2 = torch.zeros(64, device=dev)
ws = stem.workspace(x); x4 = stem.pack_input(x); w1p = stem.pack_w1(w1)
w2p, w2t, w3p, w3t = stem.pack_taps(w2, False), stem.pack_taps(w2, True), stem.pack_taps(w3, False), stem.pack_taps(w3, True)
y2, _ = stem.conv2_fwd(x4, w1p, sc1, sh1, w2p, ws)
M = B * (H // 4) * (W // 4)
dtok = (torch.randn(M, 96, device=dev) * 0.5).to(torch.bfloat16)
prm4 = torch.stack([sc2, sh2, sc2, sh2]).contiguous(); prm5 = torch.stack([sc2, sh2, sc2, sh2, sh2]).contiguous()
dy2 = stem.conv3_bwd_data(dtok, y2, prm5, w3t)
prm1 = torch.stack([sc1, sh1, sc1, sh1]).contiguous()
cases = {
    "pack_input": lambda: stem.pack_input(x),
    "conv1_stats": lambda: stem.conv1_stats(x4, w1p, ws),
    "conv2_fwd": lambda: stem.conv2_fwd(x4, w1p, sc1, sh1, w2p, ws),
    "conv3_fwd": lambda: stem.conv3_fwd(y2, sc2, sh2, w3p, b3),
    "conv3_bwd_stats": lambda: stem.conv3_bwd_stats(dtok, y2, prm4, w3t, ws),
    "conv3_bwd_data": lambda: stem.conv3_bwd_data(dtok, y2, prm5, w3t),
    "conv3_wgrad": lambda: stem.conv3_wgrad(dtok, y2, sc2, sh2, ws),
    "conv2_wgrad": lambda: stem.conv2_wgrad(x4, w1p, sc1, sh1, dy2, ws),
    "conv2_bwd": lambda: stem.conv2_bwd(x4, w1p, prm1, dy2, w2t, ws),
}
tot = 0.0
for name, fn in cases.items():
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    tot += us
    print(f"{name:18s} {us:8.1f} us")
print(f"{'total':18s} {tot:8.1f} us")
