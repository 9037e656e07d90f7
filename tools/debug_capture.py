"""Scratch: which piece of the configs[2] three-graph arrangement crashes hipStreamEndCapture in the test but not in bench.py?"""
import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from _util import TCFG
from panoswintransformerobjectdetection_amd import ops as _ops
from panoswintransformerobjectdetection_amd.detector import MiniMaskRCNN, synthetic_targets
from panoswintransformerobjectdetection_amd.dp import GradReducer
DEV = "cuda:0"
fixed = os.environ.get("FIXED_KEYS", "1") == "1"
eager_first = os.environ.get("EAGER_FIRST", "1") == "1"
which = os.environ.get("WHICH", "fwd,heads,bwd").split(",")
if os.environ.get("SET_STREAM") == "1":
    torch.cuda.set_stream(torch.cuda.Stream())
torch.manual_seed(0)
m = MiniMaskRCNN(dict(TCFG, drop_path_rate=0.0, compute_dtype=torch.bfloat16), num_classes=80).to(DEV).train()
m.backbone.init_weights(None)
if fixed:
    cache = {}
    def rand_like(t):
        n = t.numel()
        if n not in cache:
            cache[n] = torch.rand(n, generator=torch.Generator("cpu").manual_seed(1000 + n)).to(t.device)
        return cache[n].view_as(t).to(t.dtype)
    m.rand_like = rand_like
B, H, W = 2, 512, 1024
x = torch.randn(B, 3, H, W, device=DEV)
tg = synthetic_targets(B, H, W, DEV)
bb = m.backbone
heads = m.head_parameters()
if eager_first:
    outs = bb(x)
    for o in outs: o.retain_grad()
    total = sum(m.heads_loss(outs, tg, (H, W)).values())
    total.backward()
    G = [o.grad.detach().clone() for o in outs]
    for p in m.parameters(): p.grad = None
    print("eager ok", float(total), flush=True)
else:
    with torch.no_grad():
        G = [torch.zeros_like(o) for o in bb(x)]
red = GradReducer(bb, pack=True)
_ops.set_deferred_reductions(True)
gbuf = [torch.zeros_like(g) for g in G]
state = {}
def phase_fwd():
    red.zero_grad(); state["outs"] = bb(x); return state["outs"]
def phase_heads():
    feats = [o.detach().requires_grad_(True) for o in state["outs"]]
    for p in heads: p.grad = None
    ls = m.heads_loss(feats, tg, (H, W)); t = sum(ls.values()); t.backward()
    for g, f in zip(gbuf, feats): g.copy_(f.grad)
    state["losses"] = torch.stack([ls[k] for k in sorted(ls)]); return t
def phase_bwd():
    torch.autograd.backward(state["outs"], gbuf); red.pack_grads(); return gbuf[0]
fns = {"fwd": phase_fwd, "heads": phase_heads, "bwd": phase_bwd}
stream = torch.cuda.current_stream() if os.environ.get("SET_STREAM") == "1" else torch.cuda.Stream()
stream.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(stream):
    for _ in range(2):
        for k in ("fwd", "heads", "bwd"): fns[k]()
torch.cuda.current_stream().wait_stream(stream); torch.cuda.synchronize()
print("warmup ok", flush=True)
pool = None
graphs = []
for k in ("fwd", "heads", "bwd"):
    if k not in which:
        with torch.cuda.stream(stream): fns[k]()
        torch.cuda.synchronize(); print("ran eagerly", k, flush=True); continue
    g = torch.cuda.CUDAGraph()
    print("capturing", k, flush=True)
    with torch.cuda.graph(g, stream=stream, pool=pool):
        fns[k]()
    pool = pool or g.pool()
    graphs.append(g)
    print("captured", k, flush=True)
for g in graphs: g.replay()
torch.cuda.synchronize()
print("replayed ok", state["losses"].tolist(), flush=True)
