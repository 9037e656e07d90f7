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
def fb():
    red.zero_grad(); loss = sum(o.float().mean() for o in m(x)); loss.backward()
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
    return f"flat max {red.flat.abs().max().item():.4g} worst grads {worst} largest params {pw}"
for i in range(4):
    l = g(); print("replay", i, "loss", l.item(), stats(), flush=True)
    go(); print("   after opt", stats(), flush=True)
