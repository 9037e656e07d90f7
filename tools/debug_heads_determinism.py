"""scratch: are the Mask R-CNN head stand-ins deterministic on identical feature maps?  Runs heads_loss + backward several times on the
same detached backbone outputs and prints, per loss term, its value and the relative difference of its feature-map gradients to run 0."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import torch
from test_detector_gpu import _fixed_keys
from _util import TCFG
from panoswintransformerobjectdetection_amd.detector import MiniMaskRCNN, synthetic_targets

DEV = "cuda:0"
import os, time
if os.environ.get("PSWIN_HEADS_DETERMINISTIC") == "1":          # MIOpen: exclude non-deterministic solvers (miopenConvolutionAttribDeterministic)
    torch.backends.cudnn.deterministic = True
torch.manual_seed(0)
cfg = dict(TCFG, drop_path_rate=0.0, compute_dtype=torch.bfloat16)
m = MiniMaskRCNN(cfg, num_classes=80).to(DEV).train()
m.backbone.init_weights(None)
m.rand_like = _fixed_keys()
B, H, W = 2, 512, 1024
x = torch.randn(B, 3, H, W, device=DEV)
tg = synthetic_targets(B, H, W, DEV)
with torch.no_grad():
    outs = [o.clone() for o in m.backbone(x)]


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30))


ref = None
for run in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    feats = [o.detach().clone().requires_grad_(True) for o in outs]
    ls = m.heads_loss(feats, tg, (H, W))
    row = {}
    for k in sorted(ls):
        g = torch.autograd.grad(ls[k], feats, retain_graph=True, allow_unused=True)
        row[k] = (float(ls[k]), [None if t is None else t.detach().clone() for t in g])
    if ref is None:
        ref = row
    torch.cuda.synchronize()
    print("run", run, f"{(time.perf_counter() - t0) * 1e3:.1f} ms (eager, incl. the per-loss gradient passes)", flush=True)
    for k, (v, g) in row.items():
        d = [None if (a is None or b is None) else round(rel(a, b), 6) for a, b in zip(g, ref[k][1])]
        print(f"   {k:14s} {v:.9f}  (run 0: {ref[k][0]:.9f})  grad rel vs run 0: {d}", flush=True)
