"""Shared helpers for the parity tests (oracle side).  The oracle is the CHECKER, never the product."""
import os

import numpy as np
import torch
import torch.nn as nn

from detfill import det_fill_module, det_uniform

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TINY = dict(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], ape=True, drop_path_rate=0.0)
TINY_PITCH = dict(embed_dim=32, depths=[3, 2, 1, 2], num_heads=[1, 2, 4, 8], ape=True, drop_path_rate=0.0)
TCFG = dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], ape=True, drop_path_rate=0.0)
SCFG = dict(embed_dim=96, depths=[2, 2, 18, 2], num_heads=[3, 6, 12, 24], ape=True, drop_path_rate=0.0)

# parameters whose true gradient is identically zero (a bias in front of BatchNorm; the key bias, to
# which softmax is invariant): their computed gradients are rounding noise, compared with atol only.
ZERO_GRAD_KEYS = ("patch_embed.proj.0.bias", "patch_embed.proj.3.bias", "k_linear.bias")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def sub(t, n=2048):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].clone(), step


def model_inputs(shape, tag):
    return det_uniform(shape, tag + ":input", 1.0)


def loss_weights(outs, tag):
    return [det_uniform(tuple(o.shape), f"{tag}:lossw{i}", 1.0) for i, o in enumerate(outs)]


def build_filled(cls, cfg, pano, tag, train=True):
    m = cls(**cfg, pano_mode=pano)
    det_fill_module(m, tag)
    nn.Module.train(m, train)
    return m


def run_and_collect(m, shape, tag, device="cpu", subsample_out=None):
    """Mirror of oracle/gen_golden.py::run_model for any implementation of the backbone."""
    x = model_inputs(shape, tag).to(device).requires_grad_(True)
    outs = m(x)
    ws = loss_weights(outs, tag)
    loss = sum((o * w.to(device)).sum() for o, w in zip(outs, ws))
    loss.backward()
    res = {}
    for i, o in enumerate(outs):
        if subsample_out:
            res[f"out{i}_sub"] = sub(o, subsample_out)[0].cpu()
            res[f"out{i}_stats"] = torch.stack([o.mean(), o.abs().mean(), o.std()]).detach().cpu()
        else:
            res[f"out{i}"] = o.detach().cpu()
    res["dx_sub"] = sub(x.grad, 8192)[0].cpu()
    for k, p in m.named_parameters():
        if p.grad is not None:
            res[f"grad:{k}"] = sub(p.grad, 1024)[0].cpu()
            res[f"gnorm:{k}"] = p.grad.double().norm().float().cpu()
    for k, b in m.named_buffers():
        if k.endswith("running_mean") or k.endswith("running_var"):
            res[f"buf:{k}"] = b.detach().cpu().clone()
    return res


REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.json")


def record(name, **values):
    """Append measured worst-case errors to gpurun_out/parity_report.json (merged back from the GPU box): the numbers
    the tolerances in the tests are set from.  Never affects a test's verdict."""
    import json
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        data = {}
        if os.path.exists(REPORT):
            with open(REPORT) as f:
                data = json.load(f)
        data[name] = {k: (float(v) if isinstance(v, (int, float)) else v) for k, v in values.items()}
        with open(REPORT, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except (OSError, ValueError):
        pass


def rel_norm_errors(res, gold, prefixes=("grad:", "dx_sub")):
    """{key: ||got - ref|| / ||ref||} over the stored (subsampled) vectors, keys with an exactly-zero true gradient
    skipped.  This is the per-tensor bound used for the bf16 path: element-wise tolerances are meaningless for
    gradients that are sums over 10^5 tokens of bf16-rounded products."""
    out = {}
    for k in gold.files:
        if not k.startswith(prefixes) or any(z in k for z in ZERO_GRAD_KEYS):
            continue
        ref = torch.from_numpy(np.asarray(gold[k])).double()
        got = res[k].double()
        out[k] = float((got - ref).norm() / ref.norm().clamp_min(1e-30))
    return out


def compare_to_golden(res, gold, rtol, atol, grad_rtol=None, grad_atol_frac=None, skip_noise=True, report=None, dx_tol=None):
    """Outputs: allclose(rtol, atol).  Param grads and dx: |diff| <= grad_rtol*|ref| + grad_atol_frac*max|ref|
    (dx_tol = (rtol, atol_frac): its own pair for the input gradient, which passes through the train-mode BatchNorms of
    the stem and has isolated near-cancellation elements).
    report: name under which the worst err/tol ratios (outputs, gradients) are recorded (see record)."""
    grad_rtol = rtol if grad_rtol is None else grad_rtol
    grad_atol_frac = 1e-5 if grad_atol_frac is None else grad_atol_frac
    bad = []
    worst = {"out": (0.0, ""), "grad": (0.0, ""), "out_abs": (0.0, ""), "dx": (0.0, "")}
    for k in gold.files:
        if k.startswith(("gstep:", "out")) and k.endswith("_step") or k in ("dx_step", "dx_stats", "as_shimmed"):
            continue
        if k.startswith("gstep:"):
            continue
        if k not in res:
            bad.append(f"missing {k}")
            continue
        ref = torch.from_numpy(np.asarray(gold[k])).float()
        got = res[k].float()
        if got.shape != ref.shape:
            bad.append(f"shape {k}: {tuple(got.shape)} vs {tuple(ref.shape)}")
            continue
        if k.startswith(("grad:", "gnorm:", "dx_sub")):
            if skip_noise and any(z in k for z in ZERO_GRAD_KEYS):
                continue
            scale = ref.abs().max().item()
            gr, ga = dx_tol if (dx_tol is not None and k == "dx_sub") else (grad_rtol, grad_atol_frac)
            tol = gr * ref.abs() + ga * max(scale, 1e-6)
        else:
            tol = rtol * ref.abs() + atol
        err = (got - ref).abs()
        kind = "dx" if k == "dx_sub" else "grad" if k.startswith(("grad:", "gnorm:")) else "out"
        ratio = float((err / tol).max())
        if ratio > worst[kind][0]:
            worst[kind] = (ratio, k)
        if kind == "out" and float(err.max()) > worst["out_abs"][0]:
            worst["out_abs"] = (float(err.max()), k)
        if not bool((err <= tol).all()):
            i = int((err - tol).argmax())
            bad.append(f"{k}: max excess at {i}: got {got.reshape(-1)[i].item():.6g} ref {ref.reshape(-1)[i].item():.6g}")
    if report:
        record(report, out_err_over_tol=worst["out"][0], out_key=worst["out"][1], grad_err_over_tol=worst["grad"][0],
               grad_key=worst["grad"][1], dx_err_over_tol=worst["dx"][0], out_max_abs_err=worst["out_abs"][0],
               out_abs_key=worst["out_abs"][1], rel_norm_dx=rel_norm_errors(res, gold, ("dx_sub",)).get("dx_sub", 0.0),
               worst_rel_norm_grad=max(rel_norm_errors(res, gold, ("grad:",)).values(), default=0.0),
               rtol=rtol, atol=atol, grad_rtol=grad_rtol, grad_atol_frac=grad_atol_frac)
    assert not bad, "\n".join(bad[:20])
