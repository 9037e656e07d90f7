"""Every output and parameter gradient of a replayed hipGraph step against the same step run eagerly."""
import sys, torch
sys.path.insert(0, ".")
from panoswintransformerobjectdetection_amd import SimplePanoSwinTransformer
from panoswintransformerobjectdetection_amd.graph import GraphedCallable
TCFG = dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7, ape=True, drop_path_rate=0.0, pano_mode=True)
torch.manual_seed(0)
m = SimplePanoSwinTransformer(**TCFG, compute_dtype=torch.bfloat16); m.init_weights(None); m = m.cuda().train()
x = torch.randn(2, 3, 256, 512, device="cuda")
ws = [torch.randn_like(o) for o in m(x)]
def fb():
    for p in m.parameters(): p.grad = None
    outs = m(x)
    loss = sum((o.float().flatten() @ w.flatten()) for o, w in zip(outs, ws))
    loss.backward()
    return [loss] + list(outs)
ref = [t.detach().clone() for t in fb()]
gref = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
g = GraphedCallable(fb, warmup=2)
for it in range(3):
    out = g(); torch.cuda.synchronize()
    do = max(float((a.float() - b.float()).abs().max()) for a, b in zip(out, ref))
    worst = max(((float((p.grad - gref[k]).abs().max()) / (float(gref[k].abs().max()) + 1e-30), k) for k, p in m.named_parameters()))
    print("replay", it, "loss", float(out[0]), "ref", float(ref[0]), "max out diff", do, "worst rel grad diff", worst, flush=True)
