"""Repro (ROCm 7.0 / torch 2.10, MI355X): framework two-pass ("global") reductions return STALE values from the second
replay of a captured hipGraph on.

    python tools/repro_graph_stale_reduction.py

The same bf16 PatchEmbed output is reduced six ways inside one captured callable: `mean()` of the autograd output,
`mean()` / `sum()` / `double().mean()` of a detached copy (two-pass kernels that use a device semaphore whose memset is
not effective on replay), a dot product with a constant vector (single-pass GEMV) and a row-mean of row-means.  Replay 0
agrees with eager for all six; from replay 1 on the two-pass results differ while the GEMV form stays bit-identical.
That is why no such reduction is left on the product path (pswin_colsum / pswin_reduce_jobs everywhere; the bench
objective is a dot product) and why tests/test_backbone_gpu.py::test_hipgraph_replay_matches_eager exists."""
import sys

import torch

sys.path.insert(0, ".")
from panoswintransformerobjectdetection_amd import SimplePanoSwinTransformer  # noqa: E402
from panoswintransformerobjectdetection_amd.graph import GraphedCallable  # noqa: E402

TCFG = dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7, ape=True, drop_path_rate=0.0,
            pano_mode=True)
torch.manual_seed(0)
m = SimplePanoSwinTransformer(**TCFG, compute_dtype=torch.bfloat16)
m.init_weights(None)
m = m.cuda().train()
x = torch.randn(2, 3, 128, 256, device="cuda")
pe = m.patch_embed
n = 2 * 32 * 64 * 96
ONES = torch.full((n,), 1.0 / n, device="cuda")


def fb():
    for p in m.parameters():
        p.grad = None
    tok, _, _ = pe(x, torch.bfloat16)
    t2 = tok.detach().clone()
    return [tok.float().mean(), t2.mean(), t2.sum(), t2.view(-1) @ ONES, t2.double().mean(), t2.view(-1, 96).mean(0).mean()]


names = ["mean(autograd out)", "mean", "sum", "dot(ones/n)", "double().mean", "mean of row means"]
ref = [float(t) for t in fb()]
g = GraphedCallable(fb, warmup=2)
for it in range(3):
    out = g()
    torch.cuda.synchronize()
    print("replay", it, {k: (f"{float(a):.6g}", "ok" if float(a) == r else f"STALE (eager {r:.6g})")
                         for k, a, r in zip(names, out, ref)}, flush=True)
