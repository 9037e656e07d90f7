import sys, torch
sys.path.insert(0, ".")
from panoswintransformerobjectdetection_amd import SimplePanoSwinTransformer
from panoswintransformerobjectdetection_amd.dp import GradReducer
from panoswintransformerobjectdetection_amd.graph import GraphedCallable
TCFG = dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7, ape=True, drop_path_rate=0.2, pano_mode=True)
pack = sys.argv[1] == "1"
torch.manual_seed(0)
m = SimplePanoSwinTransformer(**TCFG, compute_dtype=torch.bfloat16); m.init_weights(None); m = m.cuda().train()
red = GradReducer(m, pack=pack)
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=0.05, fused=True, capturable=True)
x = torch.randn(8, 3, 512, 1024, device="cuda")
with torch.no_grad():
    LW = [torch.randn_like(o).flatten() / o.numel() for o in m(x)]
OUTS = []
def fb():
    red.zero_grad(); outs = m(x); OUTS[:] = outs; loss = sum(o.float().flatten() @ w for o, w in zip(outs, LW)); loss.backward()
    if pack: red.pack_grads()
    return loss
g = GraphedCallable(fb, warmup=2)
print("captured fb", flush=True)
go = GraphedCallable(opt.step, warmup=1, stream=g.stream)
print("captured opt", flush=True)
def stats():
    torch.cuda.synchronize()
    bad = [k for k, p in m.named_parameters() if not torch.isfinite(p).all()]
    worst = sorted(((p.grad.abs().max().item(), k) for k, p in m.named_parameters() if p.grad is not None), reverse=True)[:4]
    pw = sorted(((p.abs().max().item(), k) for k, p in m.named_parameters()), reverse=True)[:2]
    print("      outs finite", [bool(torch.isfinite(o).all()) for o in OUTS], "out absmax", [float(o.abs().max()) for o in OUTS])
    return f"flat max {red.flat.abs().max().item():.4g} worst grads {worst} largest params {pw}"
for i in range(8):
    l = g(); print("replay", i, "loss", l.item(), stats(), flush=True)
    go(); print("   after opt", stats(), flush=True)
