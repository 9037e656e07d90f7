"""Generate tests/golden/*.npz from the LIVE reference (dev container only; TEST INFRASTRUCTURE).

    python oracle/gen_golden.py            # needs /root/reference; rewrites tests/golden/

Every array below is an output of the reference's own code (imported in place by oracle/ref_loader.py)
on inputs that are reproducible from integers (oracle/detfill.py), so the fixtures stay small: weights
and inputs are regenerated, only reference OUTPUTS are stored.  Fixtures that go through the pitch
module carry ``as_shimmed = 1``: the reference cannot run that module as shipped and the
loader's per-sample shim of ``pano_rotate_image`` is part of what produced them.
"""
import os
import sys
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402
from detfill import det_fill_module, det_uniform  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
warnings.filterwarnings("ignore")

TINY = dict(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], ape=True, drop_path_rate=0.0)
TINY_PITCH = dict(embed_dim=32, depths=[3, 2, 1, 2], num_heads=[1, 2, 4, 8], ape=True, drop_path_rate=0.0)
TCFG = dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], ape=True, drop_path_rate=0.0)
# PanoSwin-S: configs/swin/mask_rcnn_swin_small_patch4_window7_mstrain_480-800_adamw_3x_coco.py:10 (depths 2,2,18,2)
SCFG = dict(embed_dim=96, depths=[2, 2, 18, 2], num_heads=[3, 6, 12, 24], ape=True, drop_path_rate=0.0)

PANO_CASES = [(128, 256), (64, 128), (32, 64), (16, 32), (13, 25), (25, 49), (50, 99), (14, 28)]
PLANAR_CASES = [(16, 32), (15, 31), (128, 256), (20, 33), (15, 25)]


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def sub(t, n=2048):
    """Strided subsample of a flattened tensor (at most n values) + the stride used."""
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].clone(), step


def gen_index_maps(hot):
    out = {"relidx_3": hot.make_relative_position_index(3), "relidx_7": hot.make_relative_position_index(7)}
    for (H, W) in PANO_CASES:
        for s in (0, 3):
            blk = hot.PanoSwinTransformerBlock(dim=32, num_heads=1, window_size=7, shift_size=s, pano_mode=True)
            ids = (torch.arange(H * W, dtype=torch.float32) + 1).view(1, H, W, 1)
            t = blk.window_transition(ids, reverse=False)
            _, SH, SW, _ = t.shape
            win = hot.window_partition(blk.pad_x(t, SH, SW), 7)
            out[f"pano_{H}x{W}_s{s}"] = (win.reshape(-1).long() - 1).to(torch.int32)
            # the reverse path (crop + reverse transition) on slot ids gives the inverse map
            Hp, Wp = blk.pad_x(t, SH, SW).shape[1:3]
            slots = torch.arange(win.numel(), dtype=torch.float32).view(-1, 7, 7, 1)
            back = hot.window_reverse(slots, 7, Hp, Wp)[:, :SH, :SW, :].contiguous()
            back = blk.window_transition(back, reverse=True)
            out[f"pano_inv_{H}x{W}_s{s}"] = back.reshape(-1).long().to(torch.int32)
    for (H, W) in PLANAR_CASES:
        for s in (0, 3):
            blk = hot.PanoSwinTransformerBlock(dim=32, num_heads=1, window_size=7, shift_size=s, pano_mode=False)
            ids = (torch.arange(H * W, dtype=torch.float32) + 1).view(1, H, W, 1)
            t = blk.window_transition(blk.pad_x(ids, H, W), reverse=False)
            out[f"planar_{H}x{W}_s{s}"] = (hot.window_partition(t, 7).reshape(-1).long() - 1).to(torch.int32)
        layer = hot.BasicLayer(dim=32, depth=2, num_heads=1, window_size=7, pano_mode=False)
        out[f"mask_{H}x{W}"] = layer._get_attention_mask(torch.zeros(1, H * W, 34), H, W).to(torch.int8)
    for (H, W) in [(5, 7), (16, 32), (13, 25), (4, 8)]:
        pm = hot.PatchMerging(dim=1)
        pm.norm, pm.reduction = nn.Identity(), nn.Identity()
        ids = (torch.arange(H * W, dtype=torch.float32) + 1).view(1, H * W, 1)
        out[f"merge_{H}x{W}"] = (pm(ids, H, W).reshape(-1, 4).long() - 1).to(torch.int32)
    save("index_maps", **out)


def gen_geometry(ns):
    hot = ns.hot
    out = {}
    for (H, W) in [(2, 4), (16, 32), (32, 64), (13, 25), (64, 128)]:
        out[f"uv_{H}x{W}"] = hot.make_uv_hw2(H, W)
    out["uv_128x256_s8"] = hot.make_uv_hw2(128, 256)[::8, ::8].contiguous()
    m = ns.SimplePanoSwinTransformer(**TINY)
    m.abs_encoder = nn.Identity()
    enc, uv = m._pano_abs_position(torch.zeros(1, 32, 16, 32))
    out["xyzuv_16x32"] = enc[0].permute(1, 2, 0).contiguous()
    # haversine on the stage-3 windows of T (16x32 tokens, pano shift 0 and 3), zero uv in pad slots
    for s in (0, 3):
        blk = hot.PanoSwinTransformerBlock(dim=32, num_heads=1, window_size=7, shift_size=s, pano_mode=True)
        uvm = hot.make_uv_hw2(16, 32)[None]
        t = blk.window_transition(uvm, reverse=False)
        win = hot.window_partition(blk.pad_x(t, t.shape[1], t.shape[2]), 7).reshape(-1, 49, 2)
        out[f"uvwin_16x32_s{s}"] = win
        out[f"hav_16x32_s{s}"] = ns.great_circle.haversine22(win, win)
    pi = np.pi
    uv1 = torch.tensor([[-77, 39], [121.489, 31.225]]) / 180 * pi
    uv2 = torch.tensor([[116.4, 39.9]] * 2) / 180 * pi
    out["city_uv1"], out["city_uv2"] = uv1, uv2
    out["city_hav22_x6400"] = ns.great_circle.haversine22(uv1, uv2) * 6400
    out["city_gc22_x6400"] = ns.great_circle.great_circle22(uv1, uv2) * 6400
    # pitch rotation (as shimmed): static two-stage resampling of a small map
    for (Hp, Wp, pr, pb) in [(14, 28, 0, 0), (21, 35, 3, 5), (7, 14, 6, 3)]:
        x = det_uniform((2, 5, Hp, Wp), f"pitch_in_{Hp}x{Wp}")
        np_uv = torch.Tensor([1.0, -0.0001]) * pi
        out[f"pitch_rot_{Hp}x{Wp}_{pr}_{pb}"] = hot.PitchAttentionModule.get_rotated(x, 7, np_uv, pr, pb)
    out["as_shimmed"] = np.int32(1)
    save("geometry", **out)


def gen_window_attention(hot):
    out = {}
    dim, heads, nW, B = 64, 2, 3, 2
    n = nW * B
    att = hot.WindowAttention(dim=dim, window_size=(7, 7), num_heads=heads, qkv_bias=True, pano_mode=True)
    att.sphere_position_alpha_table_Te = nn.Parameter(torch.zeros(169, heads))
    att.sphere_position_beta_table_Te = nn.Parameter(torch.zeros(169, heads))
    det_fill_module(att, "g4")
    feats = det_uniform((n, 49, dim), "g4:x", 1.0)
    uv = torch.stack([det_uniform((nW, 49), "g4:u", np.pi), det_uniform((nW, 49), "g4:v", np.pi / 2)], -1)
    uv[1, 40:] = 0.0                                   # a window with zero-uv padding slots
    uv = uv.repeat(B, 1, 1)
    w_out = det_uniform((n, 49, dim), "g4:wout", 1.0)
    mask3 = torch.where(det_uniform((nW, 49, 49), "g4:mask") > 0.4, torch.tensor(-100.0), torch.tensor(0.0))
    mask4 = torch.where(det_uniform((B, nW, 49, 49), "g4:mask4") > 0.4, torch.tensor(-100.0), torch.tensor(0.0))
    for name, pano, mask in (("pano", True, None), ("planar", False, None), ("planar_mask3", False, mask3),
                             ("planar_mask4", False, mask4), ("pano_mask3", True, mask3)):
        att.set_pano_mode(pano)
        att.zero_grad()
        x = torch.cat([feats, uv if pano else torch.zeros_like(uv)], -1).requires_grad_(True)
        y = att(x, mask_sOO=mask)
        (y * w_out).sum().backward()
        out[f"{name}_out"] = y
        out[f"{name}_dx"] = x.grad[..., :dim]
        for k, p in att.named_parameters():
            if p.grad is not None:
                out[f"{name}_grad_{k}"] = p.grad.clone()
    out["uv"], out["mask3"], out["mask4"] = uv, mask3.to(torch.int8), mask4.to(torch.int8)
    save("window_attention", **out)


def run_model(ns, cfg, pano, shape, tag, train=True, subsample_out=None):
    m = ns.SimplePanoSwinTransformer(**cfg, pano_mode=pano)
    for mod in m.modules():                      # D5: give alpha / beta separate storage before filling
        if hasattr(mod, "sphere_position_alpha_table_Te"):
            mod.sphere_position_alpha_table_Te = nn.Parameter(torch.zeros_like(mod.sphere_position_alpha_table_Te))
            mod.sphere_position_beta_table_Te = nn.Parameter(torch.zeros_like(mod.sphere_position_beta_table_Te))
    det_fill_module(m, tag)
    nn.Module.train(m, train)
    x = det_uniform(shape, tag + ":input", 1.0).requires_grad_(True)
    outs = m(x)
    res = {}
    loss = 0
    for i, o in enumerate(outs):
        w = det_uniform(tuple(o.shape), f"{tag}:lossw{i}", 1.0)
        loss = loss + (o * w).sum()
        if subsample_out:
            s, step = sub(o, subsample_out)
            res[f"out{i}_sub"], res[f"out{i}_step"] = s, np.int64(step)
            res[f"out{i}_stats"] = torch.stack([o.mean(), o.abs().mean(), o.std()])
        else:
            res[f"out{i}"] = o
    loss.backward()
    res["dx_sub"], res["dx_step"] = sub(x.grad, 8192)
    res["dx_stats"] = torch.stack([x.grad.mean(), x.grad.abs().mean(), x.grad.std()])
    for k, p in m.named_parameters():
        if p.grad is None:
            continue
        g, step = sub(p.grad, 1024)
        res[f"grad:{k}"], res[f"gstep:{k}"] = g, np.int64(step)
        res[f"gnorm:{k}"] = p.grad.double().norm().float()
    for k, b in m.named_buffers():
        if k.endswith("running_mean") or k.endswith("running_var"):
            res[f"buf:{k}"] = b.clone()
    return res


def main():
    """python oracle/gen_golden.py [name ...]: regenerate the named fixtures (default: all of them)."""
    ns = ref_loader.load_reference()
    assert ns is not None, "the reference is not available here"
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    jobs = {
        "index_maps": lambda: gen_index_maps(ns.hot),
        "geometry": lambda: gen_geometry(ns),
        "window_attention": lambda: gen_window_attention(ns.hot),
        "tiny_pano": lambda: save("tiny_pano", **run_model(ns, TINY, True, (2, 3, 64, 128), "tiny")),
        "tiny_planar": lambda: save("tiny_planar", **run_model(ns, TINY, False, (2, 3, 64, 128), "tiny")),
        "tiny_planar_odd": lambda: save("tiny_planar_odd", **run_model(ns, TINY, False, (2, 3, 60, 100), "tiny")),
        "tiny_pano_oddw": lambda: save("tiny_pano_oddw", **run_model(ns, TINY, True, (1, 3, 100, 196), "tiny")),
        "tiny_pitch_pano": lambda: save("tiny_pitch_pano", as_shimmed=np.int32(1),
                                        **run_model(ns, TINY_PITCH, True, (2, 3, 64, 128), "tinyp")),
        "tiny_pitch_planar": lambda: save("tiny_pitch_planar", as_shimmed=np.int32(1),
                                          **run_model(ns, TINY_PITCH, False, (2, 3, 60, 100), "tinyp")),
        # eval mode (BatchNorm running statistics, DropPath off): tiny models with all gradients, and BASELINE
        # configs[0] = PanoSwin-T forward, batch 2, 512x1024 (its backward pass is stored too: eval-mode fine-tuning)
        "tiny_pano_eval": lambda: save("tiny_pano_eval", **run_model(ns, TINY, True, (2, 3, 64, 128), "tiny", train=False)),
        "tiny_planar_eval": lambda: save("tiny_planar_eval",
                                         **run_model(ns, TINY, False, (2, 3, 60, 100), "tiny", train=False)),
        "T_512x1024_pano": lambda: save("T_512x1024_pano",
                                        **run_model(ns, TCFG, True, (2, 3, 512, 1024), "T", subsample_out=4096)),
        "T_512x1024_pano_eval": lambda: save("T_512x1024_pano_eval", **run_model(ns, TCFG, True, (2, 3, 512, 1024), "T",
                                                                                 train=False, subsample_out=4096)),
        # BASELINE configs[4] geometry: PanoSwin-S on one 1024x2048 panorama (train mode, all gradients)
        "S_1024x2048_pano": lambda: save("S_1024x2048_pano",
                                         **run_model(ns, SCFG, True, (1, 3, 1024, 2048), "S", subsample_out=4096)),
    }
    names = sys.argv[1:] or list(jobs)
    for n in names:
        jobs[n]()


if __name__ == "__main__":
    main()
