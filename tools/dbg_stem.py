import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch.nn.functional as F
from test_stem_gpu import _setup, _ref_forward, _r
B, H, W = 2, 40, 72
stem, x, w1, w2, w3, b3, sc1, sh1, sc2, sh2 = _setup(B, H, W, seed=1)
dev = x.device
g = torch.Generator().manual_seed(5)
y1, a1, y2, a2, t = _ref_forward(x, w1, w2, w3, b3, sc1, sh1, sc2, sh2)
y2k = _r(y2).permute(0, 2, 3, 1).to(torch.bfloat16).contiguous()
y2r = y2k.float().permute(0, 3, 1, 2)
M = B * (H // 4) * (W // 4)
dtok = (torch.randn(M, 96, generator=g) * 0.5).to(dev).to(torch.bfloat16)
dt_img = dtok.float().view(B, H // 4, W // 4, 96).permute(0, 3, 1, 2)
da2 = F.conv_transpose2d(dt_img, _r(w3), stride=4)
z2 = y2r * sc2[None, :, None, None] + sh2[None, :, None, None]
g2 = da2 * (z2 > 0)
k1 = (torch.rand(64, generator=g) + 0.5).to(dev)
Pv = (torch.randn(64, generator=g) * 0.05).to(dev)
Qv = (torch.randn(64, generator=g) * 0.05).to(dev)
prm5 = torch.stack([sc2, sh2, k1, Pv, Qv]).contiguous()
dy2k = stem.conv3_bwd_data(dtok, y2k, prm5, stem.pack_taps(w3, True)).float().permute(0, 3, 1, 2)
ref = k1[None, :, None, None] * g2 - Pv[None, :, None, None] * y2r - Qv[None, :, None, None]
err = (dy2k - ref).abs()
print("max err", err.max().item(), "mean", err.mean().item())
print("by channel", err.amax((0, 2, 3)))
e2 = err.amax((0, 1))
print("by tap row/col", e2.view(H // 4, 4, W // 4, 4).amax((0, 2)))
bad = (err > 0.02).nonzero()
print(bad[:10], bad.shape)
for b_, c_, y_, x_ in bad[:5].tolist():
    print(dy2k[b_, c_, y_, x_].item(), ref[b_, c_, y_, x_].item(), g2[b_, c_, y_, x_].item(), z2[b_, c_, y_, x_].item())
