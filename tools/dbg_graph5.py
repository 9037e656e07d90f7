"""Find which call breaks torch's global reductions under hipGraph replay: capture growing prefixes of the stem
forward followed by a probe reduction on a static tensor."""
import sys, torch
sys.path.insert(0, ".")
from panoswintransformerobjectdetection_amd import stem, ops, _lib
dev = "cuda:0"
B, H, W = 2, 128, 256
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 3, H, W, generator=g).to(dev)
w1 = (torch.randn(32, 3, 3, 3, generator=g) * 0.3).to(dev); w2 = (torch.randn(64, 32, 3, 3, generator=g) * 0.08).to(dev)
w3 = (torch.randn(96, 64, 4, 4, generator=g) * 0.04).to(dev); b3 = torch.zeros(96, device=dev)
bn1 = torch.nn.BatchNorm2d(32).to(dev); bn2 = torch.nn.BatchNorm2d(64).to(dev)
P = torch.randn(2 * 32 * 64 * 96, device=dev)
gam, bet = torch.ones(96, device=dev), torch.zeros(96, device=dev)
st = {}
def s_ws(): st["ws"] = stem.workspace(x)
def s_pack(): st["x4"] = stem.pack_input(x)
def s_packw(): st["w"] = stem.pack_weights(w1, w2, w3)
def s_stats(): st["s1"] = stem.conv1_stats(st["x4"], st["w"][0], st["ws"], True)
def s_fold1(): st["p1"] = stem.bn_fold_prm(st["s1"], B * H * W, bn1, None, True)
def s_conv2(): st["y2"], st["s2"] = stem.conv2_fwd(st["x4"], st["w"][0], st["p1"][0], st["p1"][1], st["w"][1], st["ws"], True)
def s_fold2(): st["p2"] = stem.bn_fold_prm(st["s2"], B * H * W, bn2, None, True)
def s_conv3(): st["tok"] = stem.conv3_fwd(st["y2"], st["p2"][0], st["p2"][1], st["w"][3], b3)
def s_ln(): st["ln"] = ops.layer_norm_gather(st["tok"].view(B, -1, 96), gam, bet, 1e-5, out_dtype=torch.float32)
steps = [s_ws, s_pack, s_packw, s_stats, s_fold1, s_conv2, s_fold2, s_conv3, s_ln]
for k in range(len(steps) + 1):
    def fb():
        for f in steps[:k]: f()
        return P.sum()
    ref = float(fb())
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fb()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        out = fb()
    vals = []
    for it in range(3):
        gr.replay(); torch.cuda.synchronize(); vals.append(float(out))
    print(k, steps[k - 1].__name__ if k else "-", "ref", f"{ref:.6g}", "replays", [f"{v:.6g}" for v in vals], flush=True)
