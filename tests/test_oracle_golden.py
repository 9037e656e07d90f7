"""The CPU oracle against the fixtures captured from the live reference (runs anywhere, no GPU)."""
import math

import numpy as np
import pytest
import torch

import panoswin_oracle as po
from _util import SCFG, TCFG, TINY, TINY_PITCH, build_filled, compare_to_golden, golden, model_inputs, run_and_collect, sub
from detfill import det_fill_module, det_uniform

PANO_CASES = [(128, 256), (64, 128), (32, 64), (16, 32), (13, 25), (25, 49), (50, 99), (14, 28)]
PLANAR_CASES = [(16, 32), (15, 31), (128, 256), (20, 33), (15, 25)]


def test_reference_known_answers():
    # HOT:105-115
    kat3 = torch.tensor([[12, 11, 10, 7, 6, 5, 2, 1, 0], [13, 12, 11, 8, 7, 6, 3, 2, 1], [14, 13, 12, 9, 8, 7, 4, 3, 2],
                         [17, 16, 15, 12, 11, 10, 7, 6, 5], [18, 17, 16, 13, 12, 11, 8, 7, 6],
                         [19, 18, 17, 14, 13, 12, 9, 8, 7], [22, 21, 20, 17, 16, 15, 12, 11, 10],
                         [23, 22, 21, 18, 17, 16, 13, 12, 11], [24, 23, 22, 19, 18, 17, 14, 13, 12]])
    assert torch.equal(po.relative_position_index(3), kat3)
    # HOT:162-171
    uv = po.uv_grid(2, 4)
    kat = torch.tensor([[[-2.3562, -0.7854], [-0.7854, -0.7854], [0.7854, -0.7854], [2.3562, -0.7854]],
                        [[-2.3562, 0.7854], [-0.7854, 0.7854], [0.7854, 0.7854], [2.3562, 0.7854]]])
    assert torch.allclose(uv, kat, atol=5e-5)
    # lzx/models/great_circle.py:108-118 (Washington / Shanghai -> Beijing, radius 6400)
    g = golden("geometry")
    d = po.haversine(torch.from_numpy(g["city_uv1"]), torch.from_numpy(g["city_uv2"])) * 6400
    assert torch.allclose(d, torch.from_numpy(g["city_hav22_x6400"]), rtol=1e-6)
    assert abs(d[0, 0].item() - 11187.2852) < 0.01 and abs(d[1, 1].item() - 1073.1840) < 0.01


def test_index_maps_bit_exact():
    g = golden("index_maps")
    assert np.array_equal(po.relative_position_index(3).numpy(), g["relidx_3"])
    assert np.array_equal(po.relative_position_index(7).numpy(), g["relidx_7"])
    for (H, W) in PANO_CASES:
        for s in (0, 3):
            m, _, _ = po.pano_window_map(H, W, s)
            assert np.array_equal(m.numpy().astype(np.int32), g[f"pano_{H}x{W}_s{s}"]), (H, W, s)
            inv = po.invert_window_map(m, H * W)
            assert np.array_equal(inv.numpy().astype(np.int32), g[f"pano_inv_{H}x{W}_s{s}"]), (H, W, s)
    for (H, W) in PLANAR_CASES:
        for s in (0, 3):
            m, _, _ = po.planar_window_map(H, W, s)
            assert np.array_equal(m.numpy().astype(np.int32), g[f"planar_{H}x{W}_s{s}"]), (H, W, s)
        mask = po.planar_attention_mask(H, W, 3)
        assert np.array_equal(mask.numpy().astype(np.int8), g[f"mask_{H}x{W}"]), (H, W)
    for (H, W) in [(5, 7), (16, 32), (13, 25), (4, 8)]:
        assert np.array_equal(po.patch_merge_map(H, W).numpy().astype(np.int32), g[f"merge_{H}x{W}"])


def test_window_transition_round_trip():
    # the reference's own check, HOT:1276-1283: reverse(forward(x)) == x for odd widths w = 2h - 1
    for h in (15, 78, 32, 94, 12, 6, 7, 45):
        w = 2 * h - 1
        for s in (0, 3):
            m, _, _ = po.pano_window_map(h, w, s)
            inv = po.invert_window_map(m, h * w)
            assert torch.equal(m[inv], torch.arange(h * w))
            assert int((m >= 0).sum()) == h * w


def test_geometry_floats():
    g = golden("geometry")
    for (H, W) in [(2, 4), (16, 32), (32, 64), (13, 25), (64, 128)]:
        assert np.array_equal(po.uv_grid(H, W).numpy(), g[f"uv_{H}x{W}"]), (H, W)
    assert np.array_equal(po.uv_grid(128, 256)[::8, ::8].numpy(), g["uv_128x256_s8"])
    assert np.allclose(po.abs_position_features(po.uv_grid(16, 32)).numpy(), g["xyzuv_16x32"], rtol=1e-6, atol=1e-7)
    for s in (0, 3):
        m, _, _ = po.pano_window_map(16, 32, s)
        uvw = po.gather_windows(po.uv_grid(16, 32).reshape(1, -1, 2), m).reshape(-1, 49, 2)
        assert np.array_equal(uvw.numpy(), g[f"uvwin_16x32_s{s}"])
        assert np.allclose(po.haversine(uvw, uvw).numpy(), g[f"hav_16x32_s{s}"], rtol=1e-6, atol=1e-7)
    np_uv = torch.Tensor([1.0, -0.0001]) * math.pi
    for (Hp, Wp, pr, pb) in [(14, 28, 0, 0), (21, 35, 3, 5), (7, 14, 6, 3)]:
        x = det_uniform((2, 5, Hp, Wp), f"pitch_in_{Hp}x{Wp}")
        rot = po.pitch_rotate_windows(x, 7, np_uv, pr, pb)                     # [B, C, nWin, 49]
        B, C = rot.shape[:2]
        rot = rot.reshape(B, C, Hp // 7, Wp // 7, 7, 7).permute(0, 1, 2, 4, 3, 5).reshape(B, C, Hp, Wp)
        assert np.allclose(rot.numpy(), g[f"pitch_rot_{Hp}x{Wp}_{pr}_{pb}"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name,pano,use_mask", [("pano", True, None), ("planar", False, None),
                                                ("planar_mask3", False, "mask3"), ("planar_mask4", False, "mask4"),
                                                ("pano_mask3", True, "mask3")])
def test_window_attention_module(name, pano, use_mask):
    g = golden("window_attention")
    dim, heads = 64, 2
    att = po.WindowAttention(dim, 7, heads)
    det_fill_module(att, "g4")
    x = det_uniform((6, 49, dim), "g4:x", 1.0).requires_grad_(True)
    uv = torch.from_numpy(g["uv"])
    mask = None if use_mask is None else torch.from_numpy(g[use_mask]).float()
    y = att(x, uv, mask, pano)
    (y * det_uniform((6, 49, dim), "g4:wout", 1.0)).sum().backward()
    assert np.allclose(y.detach().numpy(), g[f"{name}_out"], rtol=1e-5, atol=1e-6)
    assert np.allclose(x.grad.numpy(), g[f"{name}_dx"], rtol=1e-5, atol=1e-6)
    for k, p in att.named_parameters():
        key = f"{name}_grad_{k}"
        if key in g.files:
            ref = g[key]
            assert np.allclose(p.grad.numpy(), ref, rtol=1e-4, atol=1e-5 * max(1e-3, np.abs(ref).max())), k
        else:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k


@pytest.mark.parametrize("fixture,cfg,pano,shape,tag", [
    ("tiny_pano", TINY, True, (2, 3, 64, 128), "tiny"),
    ("tiny_planar", TINY, False, (2, 3, 64, 128), "tiny"),
    ("tiny_planar_odd", TINY, False, (2, 3, 60, 100), "tiny"),
    ("tiny_pano_oddw", TINY, True, (1, 3, 100, 196), "tiny"),
    ("tiny_pitch_pano", TINY_PITCH, True, (2, 3, 64, 128), "tinyp"),
    ("tiny_pitch_planar", TINY_PITCH, False, (2, 3, 60, 100), "tinyp"),
])
def test_tiny_models(fixture, cfg, pano, shape, tag):
    m = build_filled(po.SimplePanoSwinTransformerOracle, cfg, pano, tag)
    res = run_and_collect(m, shape, tag)
    compare_to_golden(res, golden(fixture), rtol=1e-5, atol=1e-6, grad_rtol=1e-4, grad_atol_frac=1e-5)


def test_T_512x1024_forward_backward():
    torch.set_num_threads(8)
    m = build_filled(po.SimplePanoSwinTransformerOracle, TCFG, True, "T")
    res = run_and_collect(m, (2, 3, 512, 1024), "T", subsample_out=4096)
    compare_to_golden(res, golden("T_512x1024_pano"), rtol=1e-4, atol=1e-5, grad_rtol=1e-3, grad_atol_frac=1e-4)


@pytest.mark.parametrize("fixture,pano,shape", [("tiny_pano_eval", True, (2, 3, 64, 128)),
                                                ("tiny_planar_eval", False, (2, 3, 60, 100))])
def test_tiny_models_eval_mode(fixture, pano, shape):
    """eval mode (HOT:981-983 train(False)): BatchNorm running statistics, DropPath off; outputs + every gradient."""
    m = build_filled(po.SimplePanoSwinTransformerOracle, TINY, pano, "tiny", train=False)
    res = run_and_collect(m, shape, "tiny")
    compare_to_golden(res, golden(fixture), rtol=1e-5, atol=1e-6, grad_rtol=1e-4, grad_atol_frac=1e-5)


def test_T_512x1024_eval_forward():
    """BASELINE.json configs[0]: PanoSwin-T forward only, batch 2, 512x1024, eval mode, against the live reference's
    captured outputs."""
    torch.set_num_threads(8)
    m = build_filled(po.SimplePanoSwinTransformerOracle, TCFG, True, "T", train=False)
    g = golden("T_512x1024_pano_eval")
    with torch.no_grad():
        outs = m(model_inputs((2, 3, 512, 1024), "T"))
    for i, o in enumerate(outs):
        ref = torch.from_numpy(g[f"out{i}_sub"])
        assert torch.allclose(sub(o, 4096)[0], ref, rtol=1e-4, atol=1e-5), i


def test_S_1024x2048_forward():
    """BASELINE.json configs[4] geometry: PanoSwin-S (depths 2-2-18-2) on one 1024x2048 panorama, train-mode forward."""
    torch.set_num_threads(8)
    m = build_filled(po.SimplePanoSwinTransformerOracle, SCFG, True, "S")
    g = golden("S_1024x2048_pano")
    with torch.no_grad():
        outs = m(model_inputs((1, 3, 1024, 2048), "S"))
    for i, o in enumerate(outs):
        ref = torch.from_numpy(g[f"out{i}_sub"])
        assert torch.allclose(sub(o, 4096)[0], ref, rtol=1e-4, atol=2e-5), i


def test_state_dict_keys_and_interface():
    m = po.SimplePanoSwinTransformerOracle(**TINY_PITCH)
    keys = set(m.state_dict().keys())
    for k in ("patch_embed.proj.0.weight", "patch_embed.proj.1.running_mean", "patch_embed.proj.6.bias",
              "patch_embed.norm.weight", "abs_encoder.weight", "layers.0.blocks.0.norm1.weight",
              "layers.0.blocks.0.attn.sphere_position_alpha_table_Te",
              "layers.0.blocks.0.attn.sphere_position_beta_table_Te",
              "layers.0.blocks.0.attn.relative_position_index_OO", "layers.0.blocks.0.attn.qkv.weight",
              "layers.0.blocks.0.attn.proj.bias", "layers.0.blocks.0.mlp.fc1.weight", "layers.0.blocks.0.mlp.fc2.bias",
              "layers.0.downsample.reduction.weight", "layers.0.downsample.norm.weight", "norm0.weight", "norm3.bias",
              "layers.0.blocks.2.np_uv", "layers.0.blocks.2.q_linear.weight", "layers.0.blocks.2.k_linear.bias",
              "layers.0.blocks.2.v_linear.weight", "layers.2.blocks.0.sphere_position_alpha_table_Te"):
        assert k in keys, k
    assert m.eval() is m and m.train() is m
    with pytest.raises(TypeError):
        m.init_weights(pretrained=3)
