"""HIP kernels (through the C ABI) against the CPU oracle and the golden fixtures.  Needs an MI355X."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import panoswin_oracle as po
from _util import golden
from detfill import det_uniform

pytestmark = pytest.mark.gpu

PANO_CASES = [(128, 256), (64, 128), (32, 64), (16, 32), (13, 25), (25, 49), (50, 99), (14, 28)]
PLANAR_CASES = [(16, 32), (15, 31), (128, 256), (20, 33), (15, 25)]


DEV = "cuda:0"


def test_library_loaded_and_abi():
    from panoswintransformerobjectdetection_amd import _lib
    lib = _lib.load()
    assert lib.pswin_version() == _lib.ABI_VERSION


def test_index_maps_bit_exact(ops):
    g = golden("index_maps")
    for (H, W) in PANO_CASES:
        for s in (0, 3):
            wmap, inv, nW = ops.window_maps(True, H, W, s, DEV)
            assert np.array_equal(wmap.cpu().numpy(), g[f"pano_{H}x{W}_s{s}"]), (H, W, s)
            assert np.array_equal(inv.cpu().numpy(), g[f"pano_inv_{H}x{W}_s{s}"]), (H, W, s)
    for (H, W) in PLANAR_CASES:
        for s in (0, 3):
            wmap, inv, nW = ops.window_maps(False, H, W, s, DEV)
            assert np.array_equal(wmap.cpu().numpy(), g[f"planar_{H}x{W}_s{s}"]), (H, W, s)
            m = wmap.cpu().long()
            assert torch.equal(m[inv.cpu().long()], torch.arange(H * W))
        mask = ops.planar_mask(H, W, 3, DEV)
        assert np.array_equal(mask.cpu().numpy().astype(np.int8), g[f"mask_{H}x{W}"]), (H, W)


def test_uv_grid_bit_exact_and_geometry(ops):
    g = golden("geometry")
    for (H, W) in [(2, 4), (16, 32), (32, 64), (13, 25), (64, 128)]:
        assert np.array_equal(ops.uv_grid(H, W, DEV).cpu().numpy().reshape(H, W, 2), g[f"uv_{H}x{W}"]), (H, W)
    assert np.array_equal(ops.uv_grid(128, 256, DEV).cpu().numpy().reshape(128, 256, 2)[::8, ::8], g["uv_128x256_s8"])
    feat = ops.abs_pos_features(16, 32, DEV).cpu().numpy().reshape(16, 32, 5)
    assert np.allclose(feat, g["xyzuv_16x32"], rtol=1e-5, atol=1e-6)
    for s in (0, 3):
        wmap, _, nW = ops.window_maps(True, 16, 32, s, DEV)
        uvw = ops.gather_uv(ops.uv_grid(16, 32, DEV), wmap).view(nW, 49, 2)
        assert np.array_equal(uvw.cpu().numpy(), g[f"uvwin_16x32_s{s}"])
        d = ops.window_dist(16, 32, s, DEV).cpu().numpy()
        assert np.allclose(d, g[f"hav_16x32_s{s}"], rtol=1e-5, atol=2e-6)
    d = ops.haversine_windows(torch.from_numpy(g["city_uv1"]).to(DEV).repeat(25, 1)[:49][None],
                              torch.from_numpy(g["city_uv2"]).to(DEV).repeat(25, 1)[:49][None])
    assert abs(d[0, 0, 0].item() * 6400 - 11187.2852) < 0.05 and abs(d[0, 1, 1].item() * 6400 - 1073.1840) < 0.05


@pytest.mark.parametrize("xdt,wdt", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16),
                                     (torch.bfloat16, torch.bfloat16)])
@pytest.mark.parametrize("H,W,C,pano,s", [(16, 32, 32, True, 3), (13, 25, 64, True, 0), (15, 31, 96, False, 3),
                                          (64, 128, 96, True, 3)])
def test_window_gather_scatter(ops, xdt, wdt, H, W, C, pano, s):
    B = 2
    x = det_uniform((B, H * W, C), "gs:x").to(xdt)
    omap = (po.pano_window_map if pano else po.planar_window_map)(H, W, s)[0]
    ref_win = po.gather_windows(x.float(), omap)
    wmap, inv, nW = ops.window_maps(pano, H, W, s, DEV)
    xd = x.to(DEV).requires_grad_(True)
    win = ops.window_gather(xd, wmap, inv, wdt)
    assert win.dtype == wdt and tuple(win.shape) == (B, nW * 49, C)
    assert torch.equal(win.float().cpu(), ref_win.to(wdt).float())           # pure copy (+ RNE cast): exact
    # scatter-add with residual and per-sample scale
    w2 = det_uniform((B, nW * 49, C), "gs:w").to(wdt)
    scale = torch.tensor([0.0, 1.25])
    ref = x.float() + scale[:, None, None] * po.scatter_windows(w2.float(), po.invert_window_map(omap, H * W))
    wd = w2.to(DEV).requires_grad_(True)
    out = ops.window_scatter_add(wd, xd, wmap, inv, scale.to(DEV))
    assert torch.allclose(out.float().cpu(), ref.to(xdt).float(), rtol=1e-6 if xdt == torch.float32 else 1e-2, atol=1e-6)
    # adjoints: gather^T = scatter
    gout = det_uniform(tuple(out.shape), "gs:g").to(xdt).to(DEV)
    out.backward(gout)
    ref_dwin = scale[:, None, None] * po.gather_windows(gout.float().cpu(), omap)
    assert torch.allclose(wd.grad.float().cpu(), ref_dwin.to(wdt).float(), rtol=1e-2 if wdt == torch.bfloat16 else 1e-6, atol=1e-6)
    xd.grad = None
    win.backward(w2.to(DEV))
    ref_dx = po.scatter_windows(w2.float(), po.invert_window_map(omap, H * W))
    assert torch.allclose(xd.grad.float().cpu(), ref_dx.to(xdt).float(), rtol=1e-2 if xdt == torch.bfloat16 else 1e-6, atol=1e-6)


@pytest.mark.parametrize("H,W", [(5, 7), (16, 32), (13, 25)])
def test_patch_merge(ops, H, W):
    B, C = 2, 32
    x = det_uniform((B, H * W, C), "pm:x")
    pm = po.patch_merge_map(H, W)
    ref = (x[:, pm.clamp(min=0).reshape(-1), :] * (pm.reshape(-1) >= 0).float()[None, :, None]).reshape(B, -1, 4 * C)
    xd = x.to(DEV).requires_grad_(True)
    out = ops.patch_merge_gather(xd, H, W)
    assert torch.equal(out.cpu(), ref)
    gout = det_uniform(tuple(out.shape), "pm:g")
    out.backward(gout.to(DEV))
    xr = x.clone().requires_grad_(True)
    ((xr[:, pm.clamp(min=0).reshape(-1), :] * (pm.reshape(-1) >= 0).float()[None, :, None]).reshape(B, -1, 4 * C) * gout).sum().backward()
    assert torch.allclose(xd.grad.cpu(), xr.grad, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("C", [32, 96, 192, 768])
@pytest.mark.parametrize("xdt,ydt", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16),
                                     (torch.bfloat16, torch.float32)])
@pytest.mark.parametrize("use_map", [False, True])
def test_layer_norm_gather(ops, C, xdt, ydt, use_map):
    B, H, W = 2, 13, 25
    x = (det_uniform((B, H * W, C), "ln:x", 2.0) + 0.3).to(xdt)
    gamma, beta = det_uniform((C,), "ln:g", 0.5, 1.0), det_uniform((C,), "ln:b", 0.5)
    xr = x.float().clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gr, br, 1e-5)
    if use_map:
        omap = po.pano_window_map(H, W, 3)[0]
        ref = po.gather_windows(ref, omap)
        wmap, inv, nW = ops.window_maps(True, H, W, 3, DEV)
    else:
        wmap = inv = None
    xd = x.to(DEV).requires_grad_(True)
    gd, bd = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
    out = ops.layer_norm_gather(xd, gd, bd, 1e-5, wmap, inv, ydt)
    assert out.dtype == ydt
    tol = dict(rtol=1e-5, atol=2e-6) if ydt == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    assert torch.allclose(out.float().cpu(), ref, **tol)
    gout = det_uniform(tuple(ref.shape), "ln:go").to(ydt)
    (ref * gout.float()).sum().backward()
    out.backward(gout.to(DEV))
    gtol = dict(rtol=1e-4, atol=1e-5) if xdt == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    assert torch.allclose(xd.grad.float().cpu(), xr.grad, **gtol)
    assert torch.allclose(gd.grad.cpu(), gr.grad, rtol=1e-4, atol=1e-4 * gr.grad.abs().max().item())
    assert torch.allclose(bd.grad.cpu(), br.grad, rtol=1e-4, atol=1e-4 * br.grad.abs().max().item())


@pytest.mark.parametrize("C", [96, 768])
@pytest.mark.parametrize("use_map", [False, True])
def test_layer_norm_gather_passthrough(ops, C, use_map):
    """x' of layer_norm_gather(..., passthrough=True) is x; the gradient sent into x' comes back added to dx (done inside
    the backward kernel): identical to the plain two-consumer graph."""
    B, H, W = 2, 13, 25
    x = det_uniform((B, H * W, C), "lnp:x", 2.0) + 0.3
    gamma, beta = det_uniform((C,), "lnp:g", 0.5, 1.0), det_uniform((C,), "lnp:b", 0.5)
    wmap, inv, nW = ops.window_maps(True, H, W, 3, DEV) if use_map else (None, None, 0)
    gx = det_uniform((B, H * W, C), "lnp:gx").to(DEV)
    res = []
    for fused in (False, True):
        xd = x.to(DEV).requires_grad_(True)
        gd, bd = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
        if fused:
            y, x2 = ops.layer_norm_gather(xd, gd, bd, 1e-5, wmap, inv, torch.bfloat16, passthrough=True)
            assert x2.data_ptr() == xd.data_ptr()
        else:
            y, x2 = ops.layer_norm_gather(xd, gd, bd, 1e-5, wmap, inv, torch.bfloat16), xd
        gy = det_uniform(tuple(y.shape), "lnp:gy").to(DEV)
        ((y.float() * gy).sum() + (x2 * gx).sum()).backward()
        res.append((y.detach(), xd.grad, gd.grad, bd.grad))
    for a, b in zip(res[0], res[1]):
        assert torch.allclose(a.float(), b.float(), rtol=1e-6, atol=1e-6)
    # only the shortcut used downstream
    xd = x.to(DEV).requires_grad_(True)
    y, x2 = ops.layer_norm_gather(xd, gamma.to(DEV), beta.to(DEV), 1e-5, wmap, inv, torch.bfloat16, passthrough=True)
    (x2 * gx).sum().backward()
    assert torch.equal(xd.grad, gx)


@pytest.mark.parametrize("C,xdt", [(96, torch.bfloat16), (96, torch.float32), (192, torch.bfloat16)])
def test_layer_norm_gather_with_added_rows_equals_the_separate_addition(ops, C, xdt):
    """pswin_ln_gather_fwd_add (PatchEmbed.norm + the absolute position rows in one pass): bitwise the LayerNorm followed by the
    broadcast addition; the rows' gradient is the batch sum of dy, the other gradients are those of the plain LayerNorm."""
    B, S = 3, 13 * 25
    x = (det_uniform((B, S, C), "lna:x", 2.0) + 0.3).to(xdt)
    gamma, beta = det_uniform((C,), "lna:g", 0.5, 1.0), det_uniform((C,), "lna:b", 0.5)
    rows = det_uniform((S, C), "lna:r", 1.5)
    gy = det_uniform((B, S, C), "lna:gy").to(DEV)
    res = []
    for fused in (False, True):
        xd = x.to(DEV).requires_grad_(True)
        gd, bd, rd = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True), rows.to(DEV).requires_grad_(True)
        if fused:
            y = ops.layer_norm_gather(xd, gd, bd, 1e-5, out_dtype=torch.float32, add_rows=rd)
        else:
            y = ops.layer_norm_gather(xd, gd, bd, 1e-5, out_dtype=torch.float32) + rd[None]
        (y * gy).sum().backward()
        res.append((y.detach(), xd.grad, gd.grad, bd.grad, rd.grad))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    with pytest.raises(Exception):
        ops.layer_norm_gather(x.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-5, out_dtype=torch.float32, add_rows=rows[:-1].to(DEV))


@pytest.mark.parametrize("H,W,C", [(16, 32, 96), (8, 16, 384), (32, 64, 192)])
@pytest.mark.parametrize("downstream", [True, False])
def test_stage_end_residual_add_with_the_output_norm_equals_the_two_separate_ops(ops, H, W, C, downstream):
    """ops.scatter_add_layer_norm_nchw (the closing MLP residual add of a stage + norm{i} as NCHW, whose backward kernel also writes the
    branch gradient bf16(scale_b dx): pswin_ln_nchw_bwd_ex) against window_scatter_add followed by layer_norm_nchw: bitwise."""
    B, S = 3, H * W
    torch.manual_seed(H + C)
    y = torch.randn(B, S, C, device=DEV).to(torch.bfloat16)
    resid = torch.randn(B, S, C, device=DEV)
    scale = torch.tensor([1.25, 0.0, 1.25], device=DEV)
    bias = torch.randn(C, device=DEV)
    gamma, beta = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV)
    g_out, g_x = torch.randn(B, C, H, W, device=DEV), torch.randn(B, S, C, device=DEV)
    assert ops.scatter_add_layer_norm_nchw_supported(y, resid)
    ident = ops.identity_map(S, DEV)
    res = []
    for fused in (False, True):
        yd, rd = y.clone().requires_grad_(True), resid.clone().requires_grad_(True)
        gd, bd = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        if fused:
            out, x = ops.scatter_add_layer_norm_nchw(yd, rd, scale, bias, gd, bd, 1e-5, H, W)
        else:
            x0 = ops.window_scatter_add(yd, rd, ident, ident, scale, bias, True)
            out, x = ops.layer_norm_nchw(x0, gd, bd, 1e-5, H, W, passthrough=True)
        loss = (out * g_out).sum() + ((x * g_x).sum() if downstream else 0.0)
        loss.backward()
        res.append((out.detach(), x.detach(), yd.grad, rd.grad, gd.grad, bd.grad))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("H,W,C", [(5, 7, 32), (16, 32, 96), (8, 16, 384), (13, 25, 192)])
@pytest.mark.parametrize("ydt", [torch.float32, torch.bfloat16])
def test_layer_norm_patch_merge(ops, H, W, C, ydt):
    B = 2
    x = det_uniform((B, H * W, C), "lnpm:x", 2.0) + 0.1
    gamma, beta = det_uniform((4 * C,), "lnpm:g", 0.5, 1.0), det_uniform((4 * C,), "lnpm:b", 0.5)
    pm = po.patch_merge_map(H, W)
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    gat = (xr[:, pm.clamp(min=0).reshape(-1), :] * (pm.reshape(-1) >= 0).float()[None, :, None]).reshape(B, -1, 4 * C)
    ref = F.layer_norm(gat, (4 * C,), gr, br, 1e-5)
    xd, gd, bd = [t.to(DEV).requires_grad_(True) for t in (x, gamma, beta)]
    out = ops.layer_norm_patch_merge(xd, gd, bd, 1e-5, H, W, ydt)
    tol = dict(rtol=1e-5, atol=2e-6) if ydt == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    assert torch.allclose(out.float().cpu(), ref, **tol)
    gout = det_uniform(tuple(ref.shape), "lnpm:go").to(ydt)
    (ref * gout.float()).sum().backward()
    out.backward(gout.to(DEV))
    assert torch.allclose(xd.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-5)
    assert torch.allclose(gd.grad.cpu(), gr.grad, rtol=1e-4, atol=1e-4 * gr.grad.abs().max().item())
    assert torch.allclose(bd.grad.cpu(), br.grad, rtol=1e-4, atol=1e-4 * br.grad.abs().max().item())


@pytest.mark.parametrize("H,W", [(14, 28), (16, 32), (4, 8)])
def test_pitch_static_resampling(ops, H, W):
    """interp_rows with the host-built tap tables == the oracle's two F.grid_sample calls."""
    from panoswintransformerobjectdetection_amd import geometry
    B, C, ws = 2, 8, 7
    np_uv = torch.Tensor([1.0, -0.0001]) * math.pi
    t = geometry.pitch_tables(H, W, ws, np_uv)
    x = det_uniform((B, H * W, C), "pitch:x").requires_grad_(True)
    img = F.pad(x.view(B, H, W, C), (0, 0, 0, t["pad_r"], 0, t["pad_b"])).permute(0, 3, 1, 2)
    ref = po.pitch_rotate_windows(img, ws, np_uv, t["pad_r"], t["pad_b"]).permute(0, 2, 3, 1).reshape(B, -1, C)
    xd = x.detach().to(DEV).requires_grad_(True)
    out = ops.interp_rows(ops.interp_rows(xd, t["idx1"].to(DEV), t["w1"].to(DEV)), t["idx2"].to(DEV), t["w2"].to(DEV))
    assert torch.allclose(out.cpu(), ref, rtol=1e-5, atol=2e-6)
    gout = det_uniform(tuple(ref.shape), "pitch:g")
    (ref * gout).sum().backward()
    out.backward(gout.to(DEV))
    assert torch.allclose(xd.grad.cpu(), x.grad, rtol=1e-4, atol=1e-5)


def _attn_case(n_rep, nW, heads, pano, mask_kind, seed):
    C = heads * 32
    n = n_rep * nW
    x = det_uniform((n * 49, 3 * C), f"att:{seed}:qkv", 1.5)
    alpha = det_uniform((169, heads), f"att:{seed}:a", 0.3)
    beta = det_uniform((169, heads), f"att:{seed}:b", 0.3)
    uv = torch.stack([det_uniform((nW, 49), f"att:{seed}:u", math.pi), det_uniform((nW, 49), f"att:{seed}:v", math.pi / 2)], -1)
    uv[0, 45:] = 0.0
    dist = po.haversine(uv, uv) if pano else None
    mask = None
    if mask_kind == 3:
        mask = torch.where(det_uniform((nW, 49, 49), f"att:{seed}:m") > 0.4, torch.tensor(-100.0), torch.tensor(0.0))
    elif mask_kind == 4:
        mask = torch.where(det_uniform((n_rep, nW, 49, 49), f"att:{seed}:m4") > 0.4, torch.tensor(-100.0), torch.tensor(0.0))
    gout = det_uniform((n * 49, C), f"att:{seed}:g", 1.0)
    return x, alpha, beta, dist, mask, gout


def _attn_oracle(x, alpha, beta, dist, mask, gout, heads, n_rep, nW):
    C = heads * 32
    x = x.clone().requires_grad_(True)
    alpha, beta = alpha.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    idx = po.relative_position_index(7).reshape(-1)
    b = beta[idx].reshape(49, 49, heads)
    if dist is not None:
        bias = dist[..., None] * alpha[idx].reshape(49, 49, heads)[None] + b
        bias = bias.repeat(n_rep, 1, 1, 1)
    else:
        bias = b[None]
    qkv = x.view(-1, 49, 3, C)
    out = po.window_attention_core(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], 32 ** -0.5, bias.permute(0, 3, 1, 2), mask,
                                   heads, 0.0, False).reshape(-1, C)
    (out * gout).sum().backward()
    return out.detach(), x.grad, alpha.grad, beta.grad


@pytest.mark.parametrize("n_rep,nW,heads,pano,mask_kind,chunks", [
    (2, 3, 2, True, 0, None), (1, 5, 1, False, 0, None), (2, 3, 3, False, 3, None), (2, 3, 2, False, 4, None),
    (3, 4, 6, True, 3, None), (8, 15, 24, True, 0, None),
    (4, 3, 2, True, 0, 1), (6, 5, 3, True, 3, 2), (3, 2, 1, False, 0, 1)])      # batch loops longer than 1 (prefetch path)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_window_attention_fwd_bwd(ops, n_rep, nW, heads, pano, mask_kind, chunks, dtype):
    x, alpha, beta, dist, mask, gout = _attn_case(n_rep, nW, heads, pano, mask_kind, f"{n_rep}{nW}{heads}")
    if dtype == torch.bfloat16:
        x = x.to(dtype).float()          # the oracle sees the same rounded inputs
        gout = gout.to(dtype).float()
    ref_out, ref_dx, ref_da, ref_db = _attn_oracle(x, alpha, beta, dist, mask, gout, heads, n_rep, nW)
    xd = x.to(DEV).to(dtype).requires_grad_(True)
    ad, bd = alpha.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
    maskd = None if mask is None else mask.reshape(-1, 49, 49).to(DEV)
    nb = nW if mask_kind != 4 else n_rep * nW
    out = ops.window_attention(xd, ad, bd, None if dist is None else dist.to(DEV), maskd, heads, 32 ** -0.5, nb,
                               chunks=chunks)
    out.backward(gout.to(DEV).to(dtype))
    # fp32: exact-f32 MFMA, differs from the oracle by summation order only.  bf16: bf16 operands (q*scale, P and
    # dS are rounded to bf16), f32 accumulation.
    rt, at = (2e-5, 2e-5) if dtype == torch.float32 else (3e-2, 3e-2)
    assert torch.allclose(out.float().cpu(), ref_out, rtol=rt, atol=at)
    gs = ref_dx.abs().max().item()
    assert torch.allclose(xd.grad.float().cpu(), ref_dx, rtol=rt * 5, atol=at * gs)
    assert torch.allclose(bd.grad.cpu(), ref_db, rtol=rt * 5, atol=at * ref_db.abs().max().item())
    if pano:
        assert torch.allclose(ad.grad.cpu(), ref_da, rtol=rt * 5, atol=at * ref_da.abs().max().item())
    else:
        assert ad.grad is None


def test_window_attention_separate_qkv(ops):
    heads, nW, n_rep = 2, 3, 2
    C = heads * 32
    x, alpha, beta, dist, _, gout = _attn_case(n_rep, nW, heads, True, 0, "sep")
    q, k, v = [x[:, i * C:(i + 1) * C].contiguous() for i in range(3)]
    fused = ops.window_attention(x.to(DEV), alpha.to(DEV), beta.to(DEV), dist.to(DEV), None, heads, 32 ** -0.5, nW)
    qd, kd, vd = [t.to(DEV).requires_grad_(True) for t in (q, k, v)]
    sep = ops.window_attention(qd, alpha.to(DEV), beta.to(DEV), dist.to(DEV), None, heads, 32 ** -0.5, nW, k=kd, v=vd)
    assert torch.equal(fused, sep)
    sep.backward(gout.to(DEV))
    _, ref_dx, _, _ = _attn_oracle(x, alpha, beta, dist, None, gout, heads, n_rep, nW)
    got = torch.cat([qd.grad, kd.grad, vd.grad], 1).cpu()
    assert torch.allclose(got, ref_dx, rtol=1e-4, atol=2e-5 * ref_dx.abs().max().item())


@pytest.mark.parametrize("C", [32, 64])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("train", [True, False])
def test_batch_norm_relu(ops, C, dt, train):
    import torch.nn as nn
    N, H, W = 2, 37, 53
    y = (det_uniform((N, C, H, W), "bn:y", 2.0) + det_uniform((1, C, 1, 1), "bn:off", 1.5)).to(dt)
    gout = det_uniform((N, C, H, W), "bn:g").to(dt)
    bn_ref = nn.BatchNorm2d(C)
    with torch.no_grad():
        bn_ref.weight.copy_(det_uniform((C,), "bn:w", 0.5, 1.0)); bn_ref.bias.copy_(det_uniform((C,), "bn:b", 0.5))
        bn_ref.running_mean.copy_(det_uniform((C,), "bn:rm", 0.3)); bn_ref.running_var.copy_(det_uniform((C,), "bn:rv", 0.3, 1.0))
    import copy
    bn_dev = copy.deepcopy(bn_ref).to(DEV)
    bn_ref.train(train); bn_dev.train(train)
    yr = y.float().clone().requires_grad_(True)
    pre = det_uniform((C,), "bn:pre", 0.7)                       # the convolution bias in front of the BN
    ref = F.relu(bn_ref(yr + pre.view(1, C, 1, 1)))
    (ref * gout.float()).sum().backward()
    yd = y.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = ops.batch_norm_relu(yd, bn_dev, train, pre.to(DEV))
    assert out.dtype == dt and out.is_contiguous(memory_format=torch.channels_last)
    out.backward(gout.to(DEV).contiguous(memory_format=torch.channels_last))
    tol = dict(rtol=1e-4, atol=1e-5) if dt == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    assert torch.allclose(out.float().cpu(), ref, **tol)
    assert torch.allclose(yd.grad.float().cpu(), yr.grad, rtol=tol["rtol"] * 5, atol=tol["atol"] * 5)
    assert torch.allclose(bn_dev.weight.grad.cpu(), bn_ref.weight.grad, rtol=1e-3 if dt == torch.float32 else 3e-2, atol=1e-3 * bn_ref.weight.grad.abs().max().item() + (0 if dt == torch.float32 else 0.3))
    assert torch.allclose(bn_dev.bias.grad.cpu(), bn_ref.bias.grad, rtol=1e-3 if dt == torch.float32 else 3e-2, atol=1e-3 * bn_ref.bias.grad.abs().max().item() + (0 if dt == torch.float32 else 0.3))
    assert torch.allclose(bn_dev.running_mean.cpu(), bn_ref.running_mean, rtol=1e-4, atol=1e-5)
    assert torch.allclose(bn_dev.running_var.cpu(), bn_ref.running_var, rtol=1e-4, atol=1e-5)
    assert int(bn_dev.num_batches_tracked) == int(bn_ref.num_batches_tracked)


@pytest.mark.parametrize("M,N,dt", [(275576 // 8, 288, torch.bfloat16), (1000, 96, torch.bfloat16), (98, 27648, torch.bfloat16),
                                    (5, 3072, torch.float32), (4097, 8, torch.float32)])
def test_colsum(ops, M, N, dt):
    x = det_uniform((M, N), "colsum").to(dt)
    ref = x.double().sum(0).float()
    got = ops.colsum(x.to(DEV)).cpu()
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-3)
    if M > 4096 and N % 24 == 0:          # columns declared zero-sum are not read: zeros there, the rest unchanged
        lo, hi = N // 3, 2 * N // 3
        part = ops.colsum(x.to(DEV), zero_cols=(lo, hi)).cpu()
        assert torch.equal(part[lo:hi], torch.zeros(hi - lo))
        assert torch.equal(part[:lo], got[:lo]) and torch.equal(part[hi:], got[hi:])


def test_grouped_reduction_many_jobs(ops):
    """pswin_reduce_jobs: 230 independent column sums (3 launches of <= 96 jobs) with every block shape (1 / 8 / 64 row
    lanes), both source dtypes, row strides larger than the summed width and a column offset, against float64 sums."""
    import random
    rnd = random.Random(7)
    jobs, want = [], []
    for i in range(230):
        dt = torch.bfloat16 if i % 3 else torch.float32
        ve = 8 if dt == torch.bfloat16 else 4
        rows = rnd.choice([1, 2, 7, 16, 17, 60, 128, 129, 700, 1536])
        cols = ve * rnd.choice([1, 3, 12, 48, 130, 1153])
        pad = ve * rnd.choice([0, 0, 2])
        off = ve * rnd.choice([0, 1]) if pad else 0
        ld = cols + pad
        src = (det_uniform((rows, ld), f"rj{i}") * 2).to(dt).to(DEV)
        jobs.append((src, off * src.element_size(), 1 if dt == torch.bfloat16 else 0, rows, cols, ld,
                     torch.full((cols,), float("nan"), device=DEV)))
        want.append(src.double()[:, off:off + cols].sum(0))
    ops._launch_reductions(jobs)
    torch.cuda.synchronize()
    for j, w in zip(jobs, want):
        assert torch.allclose(j[6].double(), w, rtol=1e-5, atol=1e-4), (j[3], j[4], j[5])


def test_table_grads_batch_matches_single_jobs(ops):
    """pswin_attn_table_grads_batch over three modules with different head counts / tile counts (with and without the
    great-circle table) == the same jobs issued one by one, and == a float64 evaluation of the definition."""
    from panoswintransformerobjectdetection_amd import _lib
    lib = _lib.load()
    idx = torch.tensor([[(i // 7 - j // 7 + 6) * 13 + (i % 7 - j % 7 + 6) for i in range(49)] for j in range(49)])  # [j][i]
    jobs, want = [], []
    for n_tiles, nb, n_dist, heads in [(150, 15, 15, 3), (40, 20, 5, 6), (9, 9, 0, 12)]:
        g = torch.zeros(n_tiles, heads, 64, 64)
        g[:, :, :49, :49] = det_uniform((n_tiles, heads, 49, 49), f"tg{heads}")
        dist = None
        if n_dist:
            dist = torch.zeros(n_dist, 64, 64)
            dist[:, :49, :49] = det_uniform((n_dist, 49, 49), f"td{heads}") + 1.0
        dbeta = torch.full((169, heads), float("nan"), device=DEV)
        dalpha = torch.full((169, heads), float("nan"), device=DEV) if n_dist else None
        ws = torch.empty(lib.pswin_attn_table_grads_workspace(heads), device=DEV)
        jobs.append((g.to(DEV), None if dist is None else dist.to(DEV), dalpha, dbeta, ws, n_tiles, nb, n_dist, heads))
        gs = g[:, :, :49, :49].double()
        wb = torch.zeros(169, heads, dtype=torch.float64)
        wb.index_add_(0, idx.flatten(), gs.sum(0).permute(1, 2, 0).reshape(49 * 49, heads))
        wa = None
        if n_dist:
            d = dist[:, :49, :49].double()[(torch.arange(n_tiles) % nb) % n_dist]            # [tile][j][i]
            wa = torch.zeros(169, heads, dtype=torch.float64)
            wa.index_add_(0, idx.flatten(), (gs * d[:, None]).sum(0).permute(1, 2, 0).reshape(49 * 49, heads))
        want.append((wa, wb))
    ops._launch_table_grads(jobs)
    torch.cuda.synchronize()
    batch = [(None if j[2] is None else j[2].clone(), j[3].clone()) for j in jobs]
    for j, (ba, bb), (wa, wb) in zip(jobs, batch, want):
        assert torch.allclose(bb.double().cpu(), wb, rtol=1e-5, atol=1e-4)
        if wa is not None:
            assert torch.allclose(ba.double().cpu(), wa, rtol=1e-5, atol=1e-4)
        j[3].fill_(float("nan"))
        ops._launch_table_grads([j])
        torch.cuda.synchronize()
        assert torch.equal(j[3], bb) and (ba is None or torch.equal(j[2], ba))


def test_rejects_cpu_and_bad_args(ops):
    from panoswintransformerobjectdetection_amd import PswinError
    with pytest.raises(PswinError):
        ops.window_gather(torch.zeros(1, 49, 32), torch.zeros(49, dtype=torch.int32), torch.zeros(49, dtype=torch.int32))
    with pytest.raises(PswinError):          # C not a multiple of 8
        ops.window_gather(torch.zeros(1, 49, 12, device=DEV), torch.zeros(49, dtype=torch.int32, device=DEV),
                          torch.zeros(49, dtype=torch.int32, device=DEV))


@pytest.mark.parametrize("M,N,dt", [(300, 384, torch.bfloat16), (77, 3072, torch.bfloat16), (129, 96, torch.float32)])
def test_bias_gelu(ops, M, N, dt):
    """gelu(y + b) and its backward (dy, db) against torch's exact GELU on the same operands; bf16: one rounding of the
    result (rtol 1e-2), the bias gradient is summed in f32 before rounding (compared with the f32 reference sum)."""
    y = det_uniform((M, N), "bg:y", 3.0).to(dt)
    b = det_uniform((N,), "bg:b", 0.5)
    gh = det_uniform((M, N), "bg:g").to(dt)
    yr, br = y.float().clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.gelu(yr + br)
    (ref * gh.float()).sum().backward()
    yd, bd = y.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    out = ops.bias_gelu(yd, bd)
    out.backward(gh.to(DEV))
    tol = dict(rtol=1e-5, atol=1e-6) if dt == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    assert torch.allclose(out.float().cpu(), ref, **tol)
    assert torch.allclose(yd.grad.float().cpu(), yr.grad, **tol)
    assert torch.allclose(bd.grad.cpu(), br.grad, rtol=1e-4, atol=1e-4 * br.grad.abs().max().item())


@pytest.mark.parametrize("use_scale", [False, True])
def test_residual_bias_pair(ops, use_scale):
    """x' + s_b (scatter(win) + bias) with the bias added by window_scatter_add and its gradient produced by the
    LayerNorm backward kernel of the same residual branch (res_bias): equal to the plain autograd graph."""
    B, H, W, C = 2, 13, 25, 96
    x = det_uniform((B, H * W, C), "rb:x", 2.0)
    gamma, beta, bias = det_uniform((C,), "rb:g", 0.5, 1.0), det_uniform((C,), "rb:b", 0.5), det_uniform((C,), "rb:bias", 0.5)
    wmap, inv, nW = ops.window_maps(True, H, W, 3, DEV)
    scale = torch.tensor([0.0, 1.25], device=DEV) if use_scale else None
    gout = det_uniform((B, H * W, C), "rb:go").to(DEV)
    wlin = det_uniform((C, C), "rb:w", 0.1).to(DEV)
    res = []
    for fused in (False, True):
        xd = x.to(DEV).requires_grad_(True)
        gd, bd, biasd = [t.to(DEV).requires_grad_(True) for t in (gamma, beta, bias)]
        if fused:
            y, x2 = ops.layer_norm_gather(xd, gd, bd, 1e-5, wmap, inv, torch.float32, passthrough=True, res_bias=biasd,
                                          res_scale=scale)
            out = ops.window_scatter_add(y @ wlin, x2, wmap, inv, scale, biasd, True)
        else:
            y = ops.layer_norm_gather(xd, gd, bd, 1e-5, wmap, inv, torch.float32)
            out = ops.window_scatter_add(y @ wlin + biasd, xd, wmap, inv, scale)
        (out * gout).sum().backward()
        res.append((out.detach(), xd.grad, gd.grad, bd.grad, biasd.grad))
    for a, b in zip(res[0], res[1]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-5 * max(1.0, float(b.abs().max())))
    # generic path: the scatter op itself returns the bias gradient when nobody else does
    xd, biasd = x.to(DEV).requires_grad_(True), bias.to(DEV).requires_grad_(True)
    win = ops.window_gather(xd, wmap, inv, torch.float32)
    out = ops.window_scatter_add(win, xd, wmap, inv, scale, biasd)
    (out * gout).sum().backward()
    assert torch.allclose(biasd.grad, res[0][4], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("C,wdt,use_scale", [(96, torch.bfloat16, True), (384, torch.bfloat16, False), (768, torch.float32, True)])
def test_scatter_add_layer_norm_equals_the_two_kernel_path(ops, C, wdt, use_scale):
    """scatter_add_layer_norm == window_scatter_add followed by layer_norm_gather(passthrough=True, res_bias=...):
    forward outputs bit for bit (same arithmetic, one pass), every gradient equal (same backward kernels)."""
    B, H, W = 2, 13, 25
    wmap, inv, nW = ops.window_maps(True, H, W, 3, DEV)
    win = det_uniform((B, nW * 49, C), "sl:w", 2.0).to(wdt)
    x = det_uniform((B, H * W, C), "sl:x", 2.0)
    gamma, beta = det_uniform((C,), "sl:g", 0.5, 1.0), det_uniform((C,), "sl:b", 0.5)
    pbias, fbias = det_uniform((C,), "sl:pb", 0.5), det_uniform((C,), "sl:fb", 0.5)
    s1 = torch.tensor([0.0, 1.25], device=DEV) if use_scale else None
    s2 = torch.tensor([1.25, 1.25], device=DEV) if use_scale else None
    gy, gx = det_uniform((B, H * W, C), "sl:gy").to(DEV), det_uniform((B, H * W, C), "sl:gx").to(DEV)
    res = []
    for fused in (False, True):
        wd, xd = win.to(DEV).requires_grad_(True), x.to(DEV).requires_grad_(True)
        gd, bd, pb, fb = [t.to(DEV).requires_grad_(True) for t in (gamma, beta, pbias, fbias)]
        if fused:
            y, x1 = ops.scatter_add_layer_norm(wd, xd, wmap, inv, s1, pb, gd, bd, 1e-5, wdt, res_bias=fb, res_scale=s2)
        else:
            x1 = ops.window_scatter_add(wd, xd, wmap, inv, s1, pb, True)
            y, x1 = ops.layer_norm_gather(x1, gd, bd, 1e-5, out_dtype=wdt, passthrough=True, res_bias=fb, res_scale=s2)
        ((y.float() * gy).sum() + (x1 * gx).sum()).backward()
        res.append((y.detach(), x1.detach(), wd.grad, xd.grad, gd.grad, bd.grad, fb.grad))
        assert pb.grad is None                                   # produced elsewhere (bias_grad_elsewhere)
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("C", [96, 192, 384, 768])
@pytest.mark.parametrize("pano,H,W,shift", [(True, 13, 25, 3), (False, 14, 21, 0), (True, 16, 32, 0)])
@pytest.mark.parametrize("use_scale", [False, True])
def test_fused_moves_of_the_layer_norm_kernels_equal_the_separate_row_movers(ops, C, pano, H, W, shift, use_scale):
    """Round 4: (1) scatter_add_layer_norm(in_pads=...): the backward kernel also writes the window gather of window_scatter_add's backward
    (pswin_ln_gather_bwd_ex) -- (2) scatter_add_layer_norm(out=...): the residual add that ends a block also runs the NEXT block's norm1
    + shift + pad + partition (pswin_scatter_add_ln_fwd_map), and its backward kernel writes the bf16 branch gradient.  Both against
    the chain of separate kernels (window_scatter_add, layer_norm_gather, window_gather): every output and every gradient bit for bit,
    zero rows in the padding slots included (planar 14 x 21 has none)."""
    B = 2
    wmap, inv, nW = ops.window_maps(pano, H, W, shift, DEV)
    pads = ops.window_pads(pano, H, W, shift, DEV)
    assert pads.numel() == nW * 49 - H * W and bool((wmap[pads.long()] < 0).all())
    S = H * W
    ident = ops.identity_map(S, DEV)
    gamma, beta = det_uniform((C,), "fm:g", 0.5, 1.0), det_uniform((C,), "fm:b", 0.5)
    pbias, fbias = det_uniform((C,), "fm:pb", 0.5), det_uniform((C,), "fm:fb", 0.5)
    s1 = torch.tensor([0.0, 1.25], device=DEV) if use_scale else None
    s2 = torch.tensor([1.25, 0.5], device=DEV) if use_scale else None
    x = det_uniform((B, S, C), "fm:x", 2.0)
    # (1) attention half: window-order input, token-order normalised output
    win = det_uniform((B, nW * 49, C), "fm:w", 2.0).to(torch.bfloat16)
    gy, gx = det_uniform((B, S, C), "fm:gy").to(DEV), det_uniform((B, S, C), "fm:gx").to(DEV)
    res = []
    for mode in ("chain", "fused"):
        wd, xd = win.to(DEV).requires_grad_(True), x.to(DEV).requires_grad_(True)
        gd, bd, pb, fb = [t.to(DEV).requires_grad_(True) for t in (gamma, beta, pbias, fbias)]
        prev = ops.LN_FUSED_MOVES
        ops.LN_FUSED_MOVES = mode == "fused"
        try:
            y, x1 = ops.scatter_add_layer_norm(wd, xd, wmap, inv, s1, pb, gd, bd, 1e-5, torch.bfloat16, res_bias=fb, res_scale=s2, in_pads=pads)
            ((y.float() * gy).sum() + (x1 * gx).sum()).backward()
        finally:
            ops.LN_FUSED_MOVES = prev
        res.append((y.detach(), x1.detach(), wd.grad, xd.grad, gd.grad, bd.grad, fb.grad))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert bool((res[1][2][:, pads.long()] == 0).all())                # the gathered gradient has zero rows in the padding slots
    # (2) the join: token-order input (the Mlp output), window-order normalised output through the NEXT block's map
    ymlp = det_uniform((B, S, C), "fm:m", 2.0).to(torch.bfloat16)
    gw = det_uniform((B, nW * 49, C), "fm:gw").to(DEV)
    res = []
    for mode in ("chain", "fused"):
        md, xd = ymlp.to(DEV).requires_grad_(True), x.to(DEV).requires_grad_(True)
        gd, bd, fb, pb = [t.to(DEV).requires_grad_(True) for t in (gamma, beta, fbias, pbias)]
        if mode == "fused":
            wn, x1 = ops.scatter_add_layer_norm(md, xd, ident, None, s2, fb, gd, bd, 1e-5, torch.bfloat16, res_bias=pb, res_scale=s1,
                                                out=(inv, nW * 49, pads))
        else:
            x1 = ops.window_scatter_add(md, xd, ident, ident, s2, fb, True)
            wn, x1 = ops.layer_norm_gather(x1, gd, bd, 1e-5, wmap, inv, torch.bfloat16, passthrough=True, res_bias=pb, res_scale=s1)
        assert wn.shape == (B, nW * 49, C)
        ((wn.float() * gw).sum() + (x1 * gx).sum()).backward()
        res.append((wn.detach(), x1.detach(), md.grad, xd.grad, gd.grad, bd.grad, pb.grad))
        assert fb.grad is None
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert bool((res[1][0][:, pads.long()] == 0).all())


@pytest.mark.parametrize("H,W,C", [(8, 16, 96), (16, 32, 192), (4, 8, 768), (16, 32, 96)])
@pytest.mark.parametrize("passthrough", [False, True])
def test_layer_norm_nchw(ops, H, W, C, passthrough):
    """LayerNorm written straight into NCHW (the output norms) against F.layer_norm + permute, forward and backward,
    with and without the residual passthrough."""
    B = 2
    x = det_uniform((B, H * W, C), "lnn:x", 2.0) + 0.2
    gamma, beta = det_uniform((C,), "lnn:g", 0.5, 1.0), det_uniform((C,), "lnn:b", 0.5)
    gy = det_uniform((B, C, H, W), "lnn:gy")
    gx = det_uniform((B, H * W, C), "lnn:gx")
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gr, br, 1e-5).view(B, H, W, C).permute(0, 3, 1, 2)
    loss = (ref * gy).sum() + ((xr * gx).sum() if passthrough else 0.0)
    loss.backward()
    xd, gd, bd = [t.to(DEV).requires_grad_(True) for t in (x, gamma, beta)]
    assert ops._lib.load().pswin_ln_nchw_supported(H * W, C)
    out = ops.layer_norm_nchw(xd, gd, bd, 1e-5, H, W, passthrough=passthrough)
    y, x2 = out if passthrough else (out, None)
    assert y.shape == (B, C, H, W) and y.is_contiguous()
    l2 = (y * gy.to(DEV)).sum() + ((x2 * gx.to(DEV)).sum() if passthrough else 0.0)
    l2.backward()
    assert torch.allclose(y.cpu(), ref, rtol=1e-5, atol=2e-6)
    assert torch.allclose(xd.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-5)
    assert torch.allclose(gd.grad.cpu(), gr.grad, rtol=1e-4, atol=1e-4 * gr.grad.abs().max().item())
    assert torch.allclose(bd.grad.cpu(), br.grad, rtol=1e-4, atol=1e-4 * br.grad.abs().max().item())


@pytest.mark.parametrize("K,N", [(96, 288), (96, 96), (96, 384), (288, 96), (384, 96), (192, 192)])
@pytest.mark.parametrize("M", [4096, 5003])
def test_skinny_gemm(ops, M, K, N):
    """Streaming GEMM against torch: forward y = x W^T + b and the transposed-weight form dx = dy W (same bf16 operands,
    f32 accumulation in another order, one bf16 rounding of the result)."""
    g = torch.Generator().manual_seed(7)
    x = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.1).to(DEV).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(DEV)
    assert ops.skinny_gemm_supported(x, N)
    y = ops.skinny_gemm(x, w, b)
    ref = x.float() @ w.float().t() + b
    assert torch.allclose(y.float(), ref, rtol=1e-2, atol=1e-2)
    # data-gradient form: contraction over the weight's rows
    dy = torch.randn(M, N, generator=g).to(DEV).to(torch.bfloat16)
    if ops._lib.load().pswin_gemm_skinny_supported(N, K):
        dx = ops.skinny_gemm(dy, w, None, transpose_w=True)
        assert dx.shape == (M, K)
        assert torch.allclose(dx.float(), dy.float() @ w.float(), rtol=1e-2, atol=2e-2)


@pytest.mark.parametrize("M", [4096, 5003])
def test_fc1_gelu_fused(ops, M):
    """fc1 + bias + GELU in the streaming GEMM and its recompute backward against torch (bf16 operands, f32 math)."""
    K, N = 96, 384
    g = torch.Generator().manual_seed(11)
    x = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.15).to(DEV)
    b = (torch.randn(N, generator=g) * 0.3).to(DEV)
    gh = torch.randn(M, N, generator=g).to(DEV).to(torch.bfloat16)
    xr = x.float().clone().requires_grad_(True)
    wr = w.to(torch.bfloat16).float().clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    ref = F.gelu(xr @ wr.t() + br)
    (ref * gh.float()).sum().backward()
    xd = x.clone().requires_grad_(True)
    wd, bd = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    assert ops.fc1_gelu_supported(xd, N)
    h = ops.fc1_gelu(xd, wd, bd)
    h.backward(gh)
    assert torch.allclose(h.float(), ref, rtol=1e-2, atol=1e-2)
    assert torch.allclose(xd.grad.float(), xr.grad, rtol=2e-2, atol=2e-2)
    rel = lambda a, r: float((a - r).norm() / r.norm())
    assert rel(wd.grad, wr.grad) < 1e-2 and rel(bd.grad, br.grad) < 1e-2


@pytest.mark.parametrize("name,pano,mask_key", [("pano", True, None), ("planar", False, None), ("planar_mask3", False, "mask3"),
                                                ("planar_mask4", False, "mask4"), ("pano_mask3", True, "mask3")])
def test_window_attention_module_against_the_reference_capture(ops, name, pano, mask_key):
    """The HIP chain qkv Linear -> pswin_attn_fwd/bwd -> proj Linear of a WindowAttention module (HOT:274-323) DIRECTLY
    against tests/golden/window_attention.npz: the live reference module's output, input gradient and every parameter
    gradient (qkv / proj weights and biases, alpha and beta tables) in 5 modes, zero-uv padding slots included."""
    from detfill import det_fill_module
    from panoswintransformerobjectdetection_amd.backbone import WindowAttention, _linear
    g = golden("window_attention")
    dim, heads, nW, B = 64, 2, 3, 2
    n = nW * B
    att = WindowAttention(dim, 7, heads)
    det_fill_module(att, "g4")
    att = att.to(DEV)
    x = det_uniform((n, 49, dim), "g4:x", 1.0).to(DEV).requires_grad_(True)
    uv = torch.from_numpy(g["uv"])[:nW].to(DEV)                      # the capture repeats the nW windows over the batch
    dist = ops.Tiles(ops.haversine_windows(uv, uv), symmetric=True) if pano else None
    mask = nb = None
    if mask_key:
        mask = ops.Tiles(torch.from_numpy(g[mask_key]).float().reshape(-1, 49, 49).to(DEV))
    nb = n if mask_key == "mask4" else nW
    qkv = _linear(x.view(-1, dim), att.qkv, torch.float32)
    o = ops.window_attention(qkv, att.sphere_position_alpha_table_Te, att.sphere_position_beta_table_Te, dist, mask,
                             heads, att.scale, nb)
    y = _linear(o, att.proj, torch.float32).view(n, 49, dim)
    (y * det_uniform((n, 49, dim), "g4:wout", 1.0).to(DEV)).sum().backward()
    assert torch.allclose(y.detach().cpu(), torch.from_numpy(g[f"{name}_out"]), rtol=2e-5, atol=2e-5)
    ref_dx = torch.from_numpy(g[f"{name}_dx"])
    assert torch.allclose(x.grad.cpu(), ref_dx, rtol=1e-4, atol=2e-5 * ref_dx.abs().max().item())
    for k, p in att.named_parameters():
        key = f"{name}_grad_{k}"
        if key in g.files:
            ref = torch.from_numpy(g[key])
            assert torch.allclose(p.grad.cpu(), ref, rtol=1e-4, atol=2e-5 * max(1e-3, ref.abs().max().item())), k
        else:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k


# ---- the per-window fused kernels: qkv -> attention -> proj (csrc/pswin_fused.hip, C = 96 / 3 heads) and qkv -> attention
# (csrc/pswin_qkvattn.hip, C = 192 / 384), bf16 ------------------------------------------------------------------------------
def _fused_case(B, nW, pano, mask_kind, seed, C=96):
    from detfill import det_fill_module
    att = po.WindowAttention(C, 7, C // 32)
    det_fill_module(att, f"fz:{seed}")
    with torch.no_grad():                       # the kernel's operands are bf16: the oracle sees the same rounded weights
        for lin in (att.qkv, att.proj):
            lin.weight.copy_(lin.weight.to(torch.bfloat16).float())
    n = B * nW
    x = det_uniform((n, 49, C), f"fz:{seed}:x", 1.0).to(torch.bfloat16).float()
    uv = torch.stack([det_uniform((nW, 49), f"fz:{seed}:u", math.pi), det_uniform((nW, 49), f"fz:{seed}:v", math.pi / 2)], -1)
    uv[0, 44:] = 0.0                            # zero-uv padding slots
    mask = None
    if mask_kind == 3:
        mask = torch.where(det_uniform((nW, 49, 49), f"fz:{seed}:m") > 0.4, torch.tensor(-100.0), torch.tensor(0.0))
    elif mask_kind == 4:
        mask = torch.where(det_uniform((B, nW, 49, 49), f"fz:{seed}:m4") > 0.4, torch.tensor(-100.0), torch.tensor(0.0))
    gout = det_uniform((n, 49, C), f"fz:{seed}:g", 1.0).to(torch.bfloat16).float()
    return att, x, uv, mask, gout


def _fused_run(ops, att_cpu, x, uv, mask, gout, B, nW, pano, mask_kind, fused, C=96):
    """The product's WindowAttention chain on the GPU in bf16: the fused kernel (C = 96: with the proj Linear; C = 192 / 384: qkv +
    attention core, then the proj GEMM) or the three-kernel chain."""
    from panoswintransformerobjectdetection_amd.backbone import WindowAttention, _linear
    heads = C // 32
    att = WindowAttention(C, 7, heads)
    att.load_state_dict(att_cpu.state_dict())
    att = att.to(DEV)
    xd = x.to(DEV).to(torch.bfloat16).view(-1, C).requires_grad_(True)
    uvd = uv.to(DEV)
    dist = ops.Tiles(ops.haversine_windows(uvd, uvd), symmetric=True) if pano else None
    mt = None if mask is None else ops.Tiles(mask.reshape(-1, 49, 49).to(DEV))
    nb = B * nW if mask_kind == 4 else nW
    if fused and C == 96:
        y = ops.window_attention_fused(xd, att, dist, mt, nb)
    elif fused:
        assert ops.window_attention_qkv_fused_supported(xd, heads)
        y = _linear(ops.window_attention_qkv_fused(xd, att, dist, mt, nb), att.proj, torch.bfloat16, use_bias=False)
    else:
        qkv = _linear(xd, att.qkv, torch.bfloat16)
        o = ops.window_attention(qkv, att.sphere_position_alpha_table_Te, att.sphere_position_beta_table_Te, dist, mt, heads,
                                 att.scale, nb)
        y = _linear(o, att.proj, torch.bfloat16, use_bias=False)
    y.backward(gout.to(DEV).to(torch.bfloat16).view(-1, C))
    grads = {k: p.grad.detach().float().cpu() for k, p in att.named_parameters() if p.grad is not None}
    return y.detach().float().cpu(), xd.grad.float().cpu(), grads


FUSED_CASES = [(2, 3, True, 0), (1, 5, False, 0), (3, 4, False, 3), (2, 3, False, 4), (2, 3, True, 3), (9, 2, True, 0),
               (8, 15, True, 0), (1, 1, True, 0)]


@pytest.mark.parametrize("B,nW,pano,mask_kind", FUSED_CASES)
def test_fused_window_attention_against_the_oracle(ops, B, nW, pano, mask_kind):
    """y = proj_nobias(attention(qkv(x))) and every gradient against the CPU oracle module on the same bf16-rounded
    weights / inputs.  Tolerance = the bf16 attention tests' (operands q, k, v, P, O rounded to bf16, f32 accumulation)."""
    att, x, uv, mask, gout = _fused_case(B, nW, pano, mask_kind, f"{B}{nW}{pano}{mask_kind}")
    xo = x.clone().requires_grad_(True)
    uvb = uv.repeat(B, 1, 1)
    yo = att(xo, uvb, mask, pano) - att.proj.bias          # the kernel leaves the proj bias to the residual scatter kernel
    (yo * gout).sum().backward()
    y, dx, grads = _fused_run(ops, att, x, uv, mask, gout, B, nW, pano, mask_kind, fused=True)
    ys = yo.abs().max().item()
    assert torch.allclose(y.view_as(yo), yo.detach(), rtol=3e-2, atol=2e-2 * ys), (y.view_as(yo) - yo).abs().max()
    gs = xo.grad.abs().max().item()
    assert torch.allclose(dx.view_as(xo.grad), xo.grad, rtol=5e-2, atol=3e-2 * gs)
    for k, p in att.named_parameters():
        if k == "proj.bias" or (k.endswith("alpha_table_Te") and not pano):
            continue
        ref = p.grad
        assert k in grads, k
        assert torch.allclose(grads[k], ref, rtol=5e-2, atol=3e-2 * ref.abs().max().item()), (k, (grads[k] - ref).abs().max())


@pytest.mark.parametrize("B,nW,pano,mask_kind", FUSED_CASES)
def test_fused_window_attention_equals_the_three_kernel_chain(ops, B, nW, pano, mask_kind):
    """Same rounding points as the unfused bf16 path (qkv, attention output rounded to bf16): outputs and gradients agree
    to a few bf16 ulps (the accumulators start from the bias instead of adding it last), and the forward-only (no
    tensors saved, V in the other MFMA orientation) mode returns the training mode's output."""
    att, x, uv, mask, gout = _fused_case(B, nW, pano, mask_kind, f"{B}{nW}{pano}{mask_kind}")
    y1, dx1, g1 = _fused_run(ops, att, x, uv, mask, gout, B, nW, pano, mask_kind, fused=True)
    y0, dx0, g0 = _fused_run(ops, att, x, uv, mask, gout, B, nW, pano, mask_kind, fused=False)
    assert torch.allclose(y1, y0, rtol=2e-2, atol=1e-2 * y0.abs().max().item())
    assert torch.allclose(dx1, dx0, rtol=2e-2, atol=1e-2 * dx0.abs().max().item())
    assert g1.keys() == g0.keys()
    for k in g0:
        assert torch.allclose(g1[k], g0[k], rtol=2e-2, atol=1e-2 * g0[k].abs().max().item()), k
    # inference mode
    from panoswintransformerobjectdetection_amd.backbone import WindowAttention
    attd = WindowAttention(96, 7, 3)
    attd.load_state_dict(att.state_dict())
    attd = attd.to(DEV)
    uvd = uv.to(DEV)
    dist = ops.Tiles(ops.haversine_windows(uvd, uvd), symmetric=True) if pano else None
    mt = None if mask is None else ops.Tiles(mask.reshape(-1, 49, 49).to(DEV))
    with torch.no_grad():
        yi = ops.window_attention_fused(x.to(DEV).to(torch.bfloat16).view(-1, 96), attd, dist, mt,
                                        B * nW if mask_kind == 4 else nW).float().cpu()
    assert torch.allclose(yi, y1, rtol=1e-2, atol=4e-3 * y1.abs().max().item())


QKV_FUSED_CASES = [(192, 2, 3, True, 0), (192, 1, 5, False, 0), (192, 3, 4, False, 3), (192, 2, 3, False, 4), (192, 9, 2, True, 0),
                   (192, 8, 50, True, 0), (384, 2, 3, True, 0), (384, 2, 3, False, 3), (384, 8, 25, True, 0), (384, 1, 1, True, 0),
                   # more images than waves: two or three items per wave and bias window (the activation ring runs on across them)
                   (384, 9, 3, False, 4), (384, 17, 2, True, 3), (192, 20, 2, True, 0),
                   # round 4, small batches: several bias windows per workgroup round (4 at C = 192, 2 at C = 384), full and ragged rounds,
                   # more rounds than one, 1 / 2 / 4 images per window
                   (192, 2, 50, True, 0), (384, 2, 50, True, 0), (192, 4, 7, False, 3), (384, 4, 9, True, 0), (192, 1, 1, True, 0),
                   (192, 1, 190, True, 0), (384, 1, 13, False, 3)]


@pytest.mark.parametrize("C,B,nW,pano,mask_kind", QKV_FUSED_CASES)
def test_fused_qkv_attention_against_the_oracle_and_the_chain(ops, C, B, nW, pano, mask_kind):
    """pswin_qkv_attn_fused_fwd (qkv Linear + attention core of one (window, head) per wave, C = 192 / 384) + the proj GEMM: the output
    and every gradient against the CPU oracle module on the same bf16-rounded weights / inputs (the bf16 attention tests' tolerance),
    and against the unfused chain qkv GEMM -> pswin_attn_fwd -> proj GEMM (same rounding points: a few bf16 ulps); the forward-only
    mode (nothing saved) returns the training mode's output bit for bit (same arithmetic, two more stores)."""
    att, x, uv, mask, gout = _fused_case(B, nW, pano, mask_kind, f"q{C}{B}{nW}{pano}{mask_kind}", C)
    xo = x.clone().requires_grad_(True)
    yo = att(xo, uv.repeat(B, 1, 1), mask, pano) - att.proj.bias
    (yo * gout).sum().backward()
    y1, dx1, g1 = _fused_run(ops, att, x, uv, mask, gout, B, nW, pano, mask_kind, True, C)
    ys, gs = yo.abs().max().item(), xo.grad.abs().max().item()
    assert torch.allclose(y1.view_as(yo), yo.detach(), rtol=3e-2, atol=2e-2 * ys), (y1.view_as(yo) - yo).abs().max()
    assert torch.allclose(dx1.view_as(xo.grad), xo.grad, rtol=5e-2, atol=3e-2 * gs)
    for k, p in att.named_parameters():
        if k == "proj.bias" or (k.endswith("alpha_table_Te") and not pano):
            continue
        assert k in g1, k
        assert torch.allclose(g1[k], p.grad, rtol=5e-2, atol=3e-2 * p.grad.abs().max().item()), (k, (g1[k] - p.grad).abs().max())
    y0, dx0, g0 = _fused_run(ops, att, x, uv, mask, gout, B, nW, pano, mask_kind, False, C)
    assert torch.allclose(y1, y0, rtol=2e-2, atol=1e-2 * y0.abs().max().item())
    assert torch.allclose(dx1, dx0, rtol=2e-2, atol=1e-2 * dx0.abs().max().item())
    assert g1.keys() == g0.keys()
    for k in g0:
        assert torch.allclose(g1[k], g0[k], rtol=2e-2, atol=1e-2 * g0[k].abs().max().item()), k
    # inference mode: the attention rows in front of proj
    from panoswintransformerobjectdetection_amd.backbone import WindowAttention
    attd = WindowAttention(C, 7, C // 32)
    attd.load_state_dict(att.state_dict())
    attd = attd.to(DEV)
    uvd = uv.to(DEV)
    dist = ops.Tiles(ops.haversine_windows(uvd, uvd), symmetric=True) if pano else None
    mt = None if mask is None else ops.Tiles(mask.reshape(-1, 49, 49).to(DEV))
    xd = x.to(DEV).to(torch.bfloat16).view(-1, C)
    nb = B * nW if mask_kind == 4 else nW
    with torch.no_grad():
        oi = ops.window_attention_qkv_fused(xd, attd, dist, mt, nb)
    ot = ops.window_attention_qkv_fused(xd.clone().requires_grad_(True), attd, dist, mt, nb)
    assert torch.equal(oi, ot.detach())


def test_fused_qkv_attention_rejects_what_it_is_not_built_for(ops):
    from panoswintransformerobjectdetection_amd import _lib
    lib = _lib.load()
    assert lib.pswin_qkv_attn_fused_supported(192, 6, _lib.BF16) == 1 and lib.pswin_qkv_attn_fused_supported(384, 12, _lib.BF16) == 1
    assert lib.pswin_qkv_attn_fused_supported(96, 3, _lib.BF16) == 0 and lib.pswin_qkv_attn_fused_supported(768, 24, _lib.BF16) == 0
    assert lib.pswin_qkv_attn_fused_supported(192, 3, _lib.BF16) == 0 and lib.pswin_qkv_attn_fused_supported(192, 6, _lib.F32) == 0
    assert not ops.window_attention_qkv_fused_supported(torch.zeros(49, 96, dtype=torch.bfloat16, device=DEV), 3)


def test_fused_window_attention_rejects_what_it_is_not_built_for(ops):
    from panoswintransformerobjectdetection_amd import PswinError, _lib
    lib = _lib.load()
    assert lib.pswin_win_attn_fused_supported(96, 3, _lib.BF16) == 1
    assert lib.pswin_win_attn_fused_supported(192, 6, _lib.BF16) == 0 and lib.pswin_win_attn_fused_supported(96, 3, _lib.F32) == 0
    x = torch.zeros(49, 192, dtype=torch.bfloat16, device=DEV)
    assert not ops.window_attention_fused_supported(x, 6)


@pytest.mark.parametrize("M,K,N,tile_m", [(19600, 384, 1152, 0), (16384, 384, 1536, 128), (4096, 3072, 768, 64), (5880, 768, 2304, 0),
                                          (200, 64, 192, 64), (333, 192, 384, 128), (74480, 192, 576, 0), (65, 128, 192, 0),
                                          (19600, 1152, 384, 96), (19600, 384, 384, 96), (100, 192, 192, 96), (5880, 768, 768, 96)])
@pytest.mark.parametrize("with_bias", [False, True])
def test_gemm_nt_against_torch(ops, M, K, N, tile_m, with_bias):
    """pswin_gemm_nt (LDS-DMA tiles, swizzled LDS, transposed-product epilogue) against an fp32 matmul of the same bf16
    operands: one bf16 rounding of the result; ragged M (partial row tiles), every supported N / K granularity."""
    torch.manual_seed(M + K + N)
    x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(torch.bfloat16)
    b = torch.randn(N, device=DEV) if with_bias else None
    assert ops.gemm_nt_supported(x, N)
    y = ops.gemm_nt(x, w, b, tile_m)
    ref = x.float() @ w.float().t()
    if with_bias:
        ref = ref + b
    assert y.shape == (M, N) and y.dtype == torch.bfloat16
    assert torch.allclose(y.float(), ref, rtol=1e-2, atol=1e-2), (y.float() - ref).abs().max()
    assert not ops.gemm_nt_supported(x, N + 64) and not ops.gemm_nt_supported(x[:, :K - 32].contiguous(), N)


@pytest.mark.parametrize("M,K,N,tile_m", [(16384, 768, 384, 0), (65536, 384, 192, 128), (4096, 1536, 768, 64), (333, 192, 384, 0), (65, 128, 192, 64)])
def test_gemm_nt_with_an_f32_result_is_the_unrounded_product(ops, M, K, N, tile_m):
    """pswin_gemm_nt_f32 (PatchMerging.reduction -> the fp32 residual stream): the same accumulators as pswin_gemm_nt, stored
    without the bf16 rounding -- its bf16 rounding IS pswin_gemm_nt's result, and it matches the fp32 matmul tightly."""
    torch.manual_seed(M + K + N)
    x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(torch.bfloat16)
    b = torch.randn(N, device=DEV)
    y32 = ops.gemm_nt(x, w, b, tile_m, out_f32=True)
    y16 = ops.gemm_nt(x, w, b, tile_m)
    assert y32.dtype == torch.float32 and y32.shape == (M, N)
    assert torch.equal(y32.to(torch.bfloat16), y16)
    ref = x.float() @ w.float().t() + b
    assert torch.allclose(y32, ref, rtol=2e-4, atol=2e-4), (y32 - ref).abs().max()


def test_linear_on_the_tiled_gemm_matches_the_library_path(ops):
    """ops.linear with the tiled HIP GEMM (forward, and the data gradient through the transposed weight copy) against the
    same layer on the library GEMMs: outputs and all three gradients within bf16 rounding."""
    import torch.nn as nn
    torch.manual_seed(0)
    lin = nn.Linear(384, 1152).to(DEV)
    lin.__dict__["_lowp"] = (lin.weight.detach().to(torch.bfloat16), lin.bias.detach().to(torch.bfloat16))
    wt = torch.empty(384, 1152, dtype=torch.bfloat16, device=DEV)
    ops.transpose_weights([(lin.__dict__["_lowp"][0], wt)])
    assert torch.equal(wt, lin.__dict__["_lowp"][0].t().contiguous())
    x = torch.randn(19600, 384, device=DEV).to(torch.bfloat16)
    g = torch.randn(19600, 1152, device=DEV).to(torch.bfloat16)

    def run(nt):
        prev, ops.GEMM_NT = ops.GEMM_NT, nt
        lin.__dict__["_lowp_t"] = wt if nt else None
        try:
            xx = x.clone().requires_grad_(True)
            lin.weight.grad = lin.bias.grad = None
            y = ops.linear(xx, lin, torch.bfloat16)
            y.backward(g)
            return y.detach().float(), xx.grad.float(), lin.weight.grad.clone(), lin.bias.grad.clone()
        finally:
            ops.GEMM_NT = prev
    assert ops.gemm_nt_tile(19600, 384, 1152) == 128
    a, b = run(True), run(False)
    for u, v in zip(a, b):
        assert torch.allclose(u, v, rtol=2e-2, atol=2e-2 * float(v.abs().max()))


def test_gemm_tn_ring_jobs_one_launch_equals_the_single_launches(ops):
    """pswin_gemm_tn_ring_jobs (round 4): many independent weight gradients in one launch per tile geometry.  Eleven jobs of all three
    geometries, ragged M, 1 .. 40 splits, f32 and bf16 slabs, with and without bias sums / zero ranges, listed in an arbitrary order
    (first_wg padding to multiples of 8, the binary search over the job table, longest-first ordering on the Python side are all in
    play): every partial slab and bias partial must equal the single launch of the same product BIT FOR BIT, and the sums an fp32 matmul."""
    import ctypes
    from panoswintransformerobjectdetection_amd import _lib
    lib = _lib.load()
    torch.manual_seed(7)
    spec = [(19600, 1152, 384, 9, True, True), (4096, 768, 3072, 2, True, False), (1470, 2304, 768, 1, False, True), (333, 192, 192, 3, False, False),
            (68894, 288, 96, 34, True, True), (65536, 96, 384, 32, True, False), (1000, 288, 96, 1, False, True), (130, 48, 96, 2, True, False),
            (5000, 192, 576, 40, True, True), (64, 192, 192, 1, False, False), (16384, 384, 1536, 8, True, True)]
    jobs, singles = [], []
    for M, N, K, sp, bf, with_bias in spec:
        dy = torch.randn(M, N, device=DEV).to(torch.bfloat16)
        x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        pdt = torch.bfloat16 if bf else torch.float32
        zc = (N // 3 // 16 * 16, 2 * (N // 3) // 16 * 16) if with_bias else None
        if with_bias:
            part1, db1 = ops.gemm_tn_ring(dy, x, sp, pdt, bias_sums=True, zero_cols=zc)
        else:
            part1, db1 = ops.gemm_tn_ring(dy, x, sp, pdt), None
        part = torch.full_like(part1, float("nan"))
        dbp = torch.full_like(db1, float("nan")) if with_bias else None
        jobs.append((dy, x, part, dbp, M, N, K, sp, zc[0] if zc else 0, zc[1] if zc else 0))
        singles.append((part1, db1))
    ops._launch_wgrads(jobs)
    torch.cuda.synchronize()
    for (dy, x, part, dbp, M, N, K, sp, zlo, zhi), (part1, db1) in zip(jobs, singles):
        assert torch.equal(part, part1), (M, N, K, sp)
        if dbp is not None:
            assert torch.equal(dbp, db1), (M, N, K, sp)
        ref = dy.float().t() @ x.float()
        assert torch.allclose(part.float().sum(0), ref, rtol=6e-3, atol=6e-3 * float(ref.abs().max()))
    # argument errors never launch
    bad = (_lib.TnJob * 1)()
    bad[0].dy, bad[0].x, bad[0].partial, bad[0].M, bad[0].N, bad[0].K, bad[0].splits, bad[0].partial_dtype = 16, 16, 16, 4096, 200, 192, 1, 0
    assert lib.pswin_gemm_tn_ring_jobs(ctypes.cast(bad, ctypes.c_void_p), 1, None) != 0
    assert lib.pswin_gemm_tn_ring_jobs(None, 1, None) != 0


def test_grouped_weight_gradients_are_bit_equal_to_immediate_ones(ops):
    """ops.queue_weight_gradient: with deferred reductions a Linear's weight (and bias) gradient is produced by the grouped end-of-pass
    launch; same split rule, kernel and summation order as the launch-where-produced mode, so the two agree bit for bit, and both with
    fp32 autograd of the same bf16 operands."""
    from panoswintransformerobjectdetection_amd.backbone import _linear
    torch.manual_seed(3)
    lins = [torch.nn.Linear(192, 576).to(DEV), torch.nn.Linear(576, 192).to(DEV), torch.nn.Linear(192, 384, bias=False).to(DEV)]
    x = torch.randn(9000, 192, device=DEV)

    def run():
        for l in lins:
            for p in l.parameters():
                p.grad = None
        h = _linear(x, lins[0], torch.bfloat16)
        h = _linear(h, lins[1], torch.bfloat16)
        h = _linear(h, lins[2], torch.bfloat16)
        h.float().square().mean().backward()
        return [p.grad.detach().clone() for l in lins for p in l.parameters()]

    assert ops.GROUPED_WGRAD and ops.gemm_tn_ring_splits(9000, 576, 192) == ops.grouped_wgrad_splits(9000) == 4
    prev = ops.set_deferred_reductions(False)
    try:
        ref = run()
        ops.set_deferred_reductions(True)
        got = run()
    finally:
        ops.set_deferred_reductions(prev)
    assert not ops._ReduceQueue.tasks
    for a, b in zip(ref, got):
        assert torch.isfinite(b).all() and torch.equal(a, b)
    # against autograd on fp32 copies of the bf16-rounded operands (loose: bf16 activations in between)
    xs = x.to(torch.bfloat16).float()
    ws = [l.weight.detach().to(torch.bfloat16).float().requires_grad_(True) for l in lins]
    h = (xs @ ws[0].t() + lins[0].bias.detach().to(torch.bfloat16).float()).to(torch.bfloat16).float()
    h = (h @ ws[1].t() + lins[1].bias.detach().to(torch.bfloat16).float()).to(torch.bfloat16).float()
    h = (h @ ws[2].t()).to(torch.bfloat16).float()
    h.square().mean().backward()
    for w, l in zip(ws, lins):
        assert float((l.weight.grad - w.grad).norm() / w.grad.norm()) < 2e-2


@pytest.mark.parametrize("M,N,K,splits", [(19600, 1152, 384, 0), (16384, 384, 1536, 0), (4096, 768, 3072, 7), (5880, 2304, 768, 0),
                                          (74480, 576, 192, 0), (74480, 192, 192, 0), (333, 192, 192, 3), (64, 192, 192, 1), (130, 384, 192, 2),
                                          (1000, 384, 384, 15), (4033, 192, 576, 63),
                                          # stage 0: the whole gradient in one tile (K = 96: qkv, proj, fc1; fc2 with K = 384, N = 96)
                                          (275576, 288, 96, 0), (275576, 96, 96, 0), (262144, 384, 96, 0), (262144, 96, 384, 0), (1000, 288, 96, 3),
                                          (777, 96, 384, 5), (130, 48, 96, 2)])
@pytest.mark.parametrize("out_bf16", [False, True])
def test_gemm_tn_ring_against_torch(ops, M, N, K, splits, out_bf16):
    """pswin_gemm_tn_ring (three-stage LDS ring, counted waits, asm transposed reads; 192 x 192 tiles: every Linear of the model
    including the stage-1 qkv / proj pair) against an fp32 matmul of the same bf16 operands: ragged M (the last slabs read past M:
    zeros; splits with fewer than three slabs; splits that start past M), 1 .. 37 slabs per split, f32 and bf16 partial slabs."""
    from panoswintransformerobjectdetection_amd import _lib
    lib = _lib.load()
    torch.manual_seed(M + N + K)
    dy = torch.randn(M, N, device=DEV).to(torch.bfloat16)
    x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    assert lib.pswin_gemm_tn_ring_supported(M, N, K) == 1
    sp = splits or lib.pswin_gemm_tn_ring_splits(M, N, K, 0)
    assert 1 <= sp <= max(1, M // 64)
    part = ops.gemm_tn_ring(dy, x, sp, torch.bfloat16 if out_bf16 else torch.float32)
    assert part.shape == (sp, N, K)
    got = part.float().sum(0)
    ref = dy.float().t() @ x.float()
    tol = 2e-3 if not out_bf16 else 6e-3                                       # bf16 slabs: one rounding per split of an O(sqrt(rows)) sum
    assert torch.allclose(got, ref, rtol=tol, atol=tol * float(ref.abs().max())), (got - ref).abs().max()
    if not out_bf16 and sp > 1:                                                # the splits partition the rows exactly
        rows = -(-(-(-M // sp)) // 64) * 64
        for s_ in (0, sp - 1):
            lo, hi = min(s_ * rows, M), min((s_ + 1) * rows, M)
            want = dy[lo:hi].float().t() @ x[lo:hi].float()
            assert torch.allclose(part[s_], want, rtol=2e-3, atol=2e-3 * float(ref.abs().max()))
    assert lib.pswin_gemm_tn_ring_supported(M, 200, K) == 0 and lib.pswin_gemm_tn_ring_supported(M, 576, 96) == 0
    # the bias gradient riding along (pswin_gemm_tn_ring_bias): same weight partials bit for bit, per-split column sums of dy, the
    # zero range written as exact zeros
    zc = (N // 3 // 16 * 16, 2 * (N // 3) // 16 * 16)
    part2, dbp = ops.gemm_tn_ring(dy, x, sp, torch.bfloat16 if out_bf16 else torch.float32, bias_sums=True, zero_cols=zc)
    assert torch.equal(part2, part) and dbp.shape == (sp, N)
    want_db = dy.float().sum(0)
    want_db[zc[0]:zc[1]] = 0
    got_db = dbp.sum(0)
    assert torch.allclose(got_db, want_db, rtol=1e-4, atol=1e-4 * float(dy.float().abs().sum(0).max())), (got_db - want_db).abs().max()
    assert bool((dbp[:, zc[0]:zc[1]] == 0).all())
    if sp > 1:
        rows = -(-(-(-M // sp)) // 64) * 64
        lo, hi = min((sp - 1) * rows, M), M
        w_last = dy[lo:hi].float().sum(0)
        w_last[zc[0]:zc[1]] = 0
        assert torch.allclose(dbp[sp - 1], w_last, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("large_first", [False, True])
def test_deferred_reductions_with_a_linear_applied_to_a_large_and_a_small_input(ops, large_first):
    """One nn.Linear used twice in a pass, once on enough rows for a split weight gradient (postponed to the end of the pass) and once
    on a handful (one launch, returned to autograd at once): whichever comes second, autograd must never add an unfilled tensor
    (the immediate path behind a postponed gradient of the same weight)."""
    import torch.nn as nn
    torch.manual_seed(5)
    lin = nn.Linear(192, 192).to(DEV)
    lin.__dict__["_lowp"] = (lin.weight.detach().to(torch.bfloat16), lin.bias.detach().to(torch.bfloat16))
    xl = torch.randn(8192, 192, device=DEV).to(torch.bfloat16)
    xs = torch.randn(64, 192, device=DEV).to(torch.bfloat16)

    def run():
        lin.weight.grad = lin.bias.grad = None
        a, b = (xl, xs) if large_first else (xs, xl)
        y = ops.linear(a, lin, torch.bfloat16).float().sum() + 2.0 * ops.linear(b, lin, torch.bfloat16).float().sum()
        y.backward()
        torch.cuda.synchronize()
        return lin.weight.grad.clone(), lin.bias.grad.clone()

    want = run()
    prev = ops.set_deferred_reductions(True)
    try:
        got = run()
    finally:
        ops.set_deferred_reductions(prev)
    for g, w in zip(got, want):
        assert torch.isfinite(g).all()
        assert torch.allclose(g, w, rtol=1e-3, atol=1e-3 * float(w.abs().max())), float((g - w).abs().max())


@pytest.mark.parametrize("M", [4096, 20000, 262144 + 48])
def test_stage0_mlp_backward_in_one_pass(ops, M):
    """pswin_mlp0_bwd: g = (dy W2) * gelu'(x W1^T + b1) and its column sums against fp32 torch on the same bf16 operands (ragged M:
    rows past M must not enter the sums), then the autograd node ops.mlp0_fused against the two-node chain it replaces
    (pswin_fc1_gelu_fwd/bwd + the fc2 Linear), whose dh is rounded to bf16 on its way through memory."""
    import torch.nn as nn
    from panoswintransformerobjectdetection_amd import _lib
    lib = _lib.load()
    torch.manual_seed(M)
    C, Hd = 96, 384
    assert lib.pswin_mlp0_bwd_supported(C, Hd) == 1 and lib.pswin_mlp0_bwd_supported(192, 768) == 0
    fc1, fc2 = nn.Linear(C, Hd).to(DEV), nn.Linear(Hd, C).to(DEV)
    x = torch.randn(M, C, device=DEV).to(torch.bfloat16)
    dy = (torch.randn(M, C, device=DEV) * 0.1).to(torch.bfloat16)
    w1b, w2b = fc1.weight.detach().to(torch.bfloat16).contiguous(), fc2.weight.detach().to(torch.bfloat16).contiguous()
    b1 = fc1.bias.detach().float().contiguous()
    rows = lib.pswin_mlp0_bwd_partial_rows(M)
    assert 1 <= rows <= 256
    g = torch.empty(M, Hd, dtype=torch.bfloat16, device=DEV)
    ws = torch.empty(rows, Hd, dtype=torch.float32, device=DEV)
    db = torch.empty(Hd, dtype=torch.float32, device=DEV)
    ops.call("pswin_mlp0_bwd", x, ops.ptr(x), ops.ptr(w1b), ops.ptr(b1), ops.ptr(dy), ops.ptr(w2b), ops.ptr(g), ops.ptr(db), ops.ptr(ws), M, C, Hd)
    pre = x.float() @ w1b.float().t() + b1
    dh = dy.float() @ w2b.float()
    cdf = 0.5 * (1 + torch.erf(pre / math.sqrt(2.0)))
    want = dh * (cdf + pre * torch.exp(-0.5 * pre * pre) / math.sqrt(2 * math.pi))
    scale = float(want.abs().max())
    assert torch.allclose(g.float(), want, rtol=1e-2, atol=2e-3 * scale), float((g.float() - want).abs().max())
    assert torch.allclose(ws.sum(0), db, rtol=1e-5, atol=1e-4)
    assert torch.allclose(db, g.float().sum(0), rtol=2e-3, atol=2e-3 * float(g.float().abs().sum(0).max()))
    g2 = torch.empty_like(g)                                                   # without column sums (workspace = NULL): the same rows
    ops.call("pswin_mlp0_bwd", x, ops.ptr(x), ops.ptr(w1b), ops.ptr(b1), ops.ptr(dy), ops.ptr(w2b), ops.ptr(g2), None, None, M, C, Hd)
    assert torch.equal(g2, g)
    # the autograd node against the chain
    for lin in (fc1, fc2):
        lin.__dict__["_lowp"] = (lin.weight.detach().to(torch.bfloat16), lin.bias.detach().to(torch.bfloat16))
    xa = x.clone().requires_grad_(True)
    ya = ops.mlp0_fused(xa, fc1, fc2)
    ya.backward(dy)
    ga = [xa.grad.float(), fc1.weight.grad.clone(), fc1.bias.grad.clone(), fc2.weight.grad.clone()]
    for p_ in (fc1.weight, fc1.bias, fc2.weight):
        p_.grad = None
    xb = x.clone().requires_grad_(True)
    hb = ops.fc1_gelu(xb, fc1.weight, fc1.bias, fc1.__dict__["_lowp"][0])
    yb = ops.linear(hb, fc2, torch.bfloat16, use_bias=False)
    # forward: one pass (pswin_mlp0_fwd) against fc1 + GELU kernel -> streaming fc2: same bf16 h, y up to the order of the f32 sums
    ya_d, yb_d = ya.detach().float(), yb.detach().float()
    assert torch.allclose(ya_d, yb_d, rtol=1e-2, atol=2e-3 * float(yb_d.abs().max())), float((ya_d - yb_d).abs().max())
    h_ref = torch.nn.functional.gelu(x.float() @ w1b.float().t() + b1)
    y_ref = h_ref.to(torch.bfloat16).float() @ w2b.float().t()
    assert torch.allclose(ya_d, y_ref, rtol=1e-2, atol=4e-3 * float(y_ref.abs().max())), float((ya_d - y_ref).abs().max())
    hh, yy = torch.empty(M, Hd, dtype=torch.bfloat16, device=DEV), torch.empty(M, C, dtype=torch.bfloat16, device=DEV)
    ops.call("pswin_mlp0_fwd", x, ops.ptr(x), ops.ptr(w1b), ops.ptr(b1), ops.ptr(w2b), ops.ptr(hh), ops.ptr(yy), M, C, Hd)
    assert torch.equal(hh, hb.detach()) and torch.equal(yy, ya.detach())       # h: the very values the fc1 + GELU kernel stores
    yb.backward(dy)
    gb = [xb.grad.float(), fc1.weight.grad, fc1.bias.grad, fc2.weight.grad]
    for a_, b_ in zip(ga, gb):
        assert torch.allclose(a_, b_, rtol=2e-2, atol=1e-2 * float(b_.abs().max())), float((a_ - b_).abs().max())


@pytest.mark.parametrize("M,C,tile", [(16384, 384, 128), (4096, 768, 64), (333, 192, 64), (19600, 384, 128)])
def test_gelu_backward_fused_into_the_fc2_data_gradient(ops, M, C, tile):
    """pswin_gemm_nt_gelu_bwd: (dy . W2) * gelu'(pre + b1) and the per-tile column sums, against fp32 torch on the same bf16
    operands (ragged M: rows past M must not enter the column sums); then the autograd node ops.bias_gelu_linear against the
    two-kernel chain bias_gelu + linear."""
    import torch.nn as nn
    from panoswintransformerobjectdetection_amd import _lib
    lib = _lib.load()
    torch.manual_seed(M + C)
    N = 4 * C
    dy = torch.randn(M, C, device=DEV).to(torch.bfloat16)
    w2 = (torch.randn(C, N, device=DEV) / math.sqrt(N)).to(torch.bfloat16)              # fc2.weight [C, 4C]
    pre = torch.randn(M, N, device=DEV).to(torch.bfloat16)
    b1 = torch.randn(N, device=DEV) * 0.1
    wt = w2.t().contiguous()                                                            # [4C, C]: the kernel's "weight"
    rows = lib.pswin_gemm_nt_partial_rows(M, tile)
    dpre = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ws = torch.empty(rows, N, dtype=torch.float32, device=DEV)
    ops.call("pswin_gemm_nt_gelu_bwd", dy, ops.ptr(dy), ops.ptr(wt), ops.ptr(pre), ops.ptr(b1), ops.ptr(dpre), ops.ptr(ws), M, C, N, tile)
    z = (pre.float() + b1).requires_grad_(True)
    F.gelu(z).backward(dy.float() @ w2.float())
    ref = z.grad
    assert torch.allclose(dpre.float(), ref, rtol=2e-2, atol=2e-2)
    assert torch.allclose(ws.sum(0), ref.sum(0), rtol=1e-2, atol=2e-2 * float(ref.sum(0).abs().max()))
    # the autograd node against the two-kernel chain
    fc2 = nn.Linear(N, C).to(DEV)
    with torch.no_grad():
        fc2.weight.copy_(w2.float())
    fc2.__dict__["_lowp"] = (w2, None)
    fc2.__dict__["_lowp_t"] = wt
    bias1 = nn.Parameter(b1.clone())
    g = torch.randn(M, C, device=DEV).to(torch.bfloat16)

    def run(fused):
        y = pre.clone().requires_grad_(True)
        bias1.grad = fc2.weight.grad = None
        out = ops.bias_gelu_linear(y, bias1, fc2) if fused else ops.linear(ops.bias_gelu(y, bias1), fc2, torch.bfloat16, use_bias=False)
        out.backward(g)
        return out.detach().float(), y.grad.float(), bias1.grad.clone(), fc2.weight.grad.clone()
    a, b = run(True), run(False)
    for u, v in zip(a, b):
        assert torch.allclose(u, v, rtol=2e-2, atol=2e-2 * float(v.abs().max())), (u - v).abs().max()


@pytest.mark.parametrize("M,C,tile", [(16384, 192, 128), (4096, 384, 64), (333, 192, 64), (1024, 768, 64)])
def test_fc1_with_the_gelu_in_its_epilogue(ops, M, C, tile):
    """pswin_gemm_nt_gelu_fwd writes the pre-activation (bit-identical to the tiled GEMM) and gelu(pre + b1) (the streaming bias +
    GELU kernel's value from the same rounded pre-activation, within one bf16 ulp); then the single-node MLP ops.mlp_fused against
    the chain linear -> bias_gelu_linear (output and every gradient)."""
    import torch.nn as nn
    torch.manual_seed(M + C)
    N = 4 * C
    x = torch.randn(M, C, device=DEV).to(torch.bfloat16)
    w1 = (torch.randn(N, C, device=DEV) / math.sqrt(C)).to(torch.bfloat16)
    b1 = torch.randn(N, device=DEV) * 0.1
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    h = torch.empty_like(pre)
    ops.call("pswin_gemm_nt_gelu_fwd", x, ops.ptr(x), ops.ptr(w1), ops.ptr(b1), ops.ptr(pre), ops.ptr(h), M, C, N, tile)
    pre_ref = ops.gemm_nt(x, w1, None, tile)
    h_ref = ops.bias_gelu(pre_ref, b1)
    assert torch.equal(pre, pre_ref)
    # the epilogue evaluates the GELU two elements at a time on packed-f32 instructions (folded constants, another product
    # order): the same value to f32 rounding, so at most one bf16 ulp apart from the streaming kernel, and that rarely
    d = (h.float() - h_ref.float()).abs()
    assert bool((d <= h_ref.float().abs() * 2.0 ** -7 + 1e-30).all()), d.max()
    assert float((d > 0).float().mean()) < 1e-3
    fp = F.gelu(x.float() @ w1.float().t() + b1)
    assert torch.allclose(h.float(), fp, rtol=2e-2, atol=2e-2)

    fc1, fc2 = nn.Linear(C, N).to(DEV), nn.Linear(N, C).to(DEV)
    with torch.no_grad():
        fc1.weight.copy_(w1.float()); fc1.bias.copy_(b1)
    for lin in (fc1, fc2):
        wb = lin.weight.detach().to(torch.bfloat16)
        lin.__dict__["_lowp"] = (wb, None)
        lin.__dict__["_lowp_t"] = wb.t().contiguous()
    g = torch.randn(M, C, device=DEV).to(torch.bfloat16)
    assert ops.mlp_fused_supported(x, N)

    def run(fused):
        xx = x.clone().requires_grad_(True)
        for p in (*fc1.parameters(), *fc2.parameters()):
            p.grad = None
        if fused:
            out = ops.mlp_fused(xx, fc1, fc2)
        else:
            out = ops.bias_gelu_linear(ops.linear(xx, fc1, torch.bfloat16, use_bias=False), fc1.bias, fc2)
        out.backward(g)
        return out.detach().float(), xx.grad.float(), fc1.weight.grad.clone(), fc1.bias.grad.clone(), fc2.weight.grad.clone()
    a, b = run(True), run(False)
    for u, v in zip(a, b):
        assert torch.allclose(u, v, rtol=2e-2, atol=2e-2 * float(v.abs().max())), (u - v).abs().max()
    with torch.no_grad():
        assert torch.equal(ops.mlp_fused(x, fc1, fc2), a[0].to(torch.bfloat16))


def test_flat_adamw_matches_torch_adamw(ops):
    """optim.FlatAdamW (pswin_adamw_flat: one launch over the flat parameter buffer, bf16 copy of the new weights in the same pass)
    against torch.optim.AdamW on the same gradients for 6 steps: parameters and both moments to f32 rounding, the bf16 shadow
    exactly bf16(parameters)."""
    from panoswintransformerobjectdetection_amd.optim import FlatAdamW
    torch.manual_seed(7)
    n = 4 * 100003
    p0 = torch.randn(n, device=DEV)
    ref = torch.nn.Parameter(p0.clone())
    mine = torch.nn.Parameter(p0.clone())
    kw = dict(lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    o_ref = torch.optim.AdamW([ref], **kw)
    o_mine = FlatAdamW(mine, **kw)
    shadow = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    o_mine.lowp = shadow
    for step in range(6):
        g = torch.randn(n, device=DEV) * (0.1 + step)
        ref.grad, mine.grad = g.clone(), g.clone()
        o_ref.step()
        o_mine.step()
        assert torch.allclose(mine.data, ref.data, rtol=2e-6, atol=2e-7), (step, (mine.data - ref.data).abs().max())
        assert torch.equal(shadow, mine.data.to(torch.bfloat16))
    st = o_ref.state[ref]
    assert torch.allclose(o_mine.exp_avg, st["exp_avg"], rtol=1e-5, atol=1e-6)            # (sums with cancellation: absolute bound)
    assert torch.allclose(o_mine.exp_avg_sq, st["exp_avg_sq"], rtol=1e-5, atol=1e-8)
    assert float(o_mine.step_t) == 6.0


def test_flat_adamw_state_dict_round_trip_with_torch_adamw(ops):
    """The state of optim.FlatAdamW sits under torch.optim.AdamW's keys: a state dict saved by one continues in the other."""
    from panoswintransformerobjectdetection_amd.optim import FlatAdamW
    torch.manual_seed(11)
    n = 4096
    kw = dict(lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    a = torch.nn.Parameter(torch.randn(n, device=DEV))
    b = torch.nn.Parameter(a.data.clone())
    oa, ob = FlatAdamW(a, **kw), torch.optim.AdamW([b], capturable=True, **kw)
    gs = [torch.randn(n, device=DEV) for _ in range(4)]
    for g in gs[:2]:
        a.grad, b.grad = g.clone(), g.clone()
        oa.step(); ob.step()
    # swap: torch continues from the HIP optimizer's state and vice versa
    sa, sb = oa.state_dict(), ob.state_dict()
    oa2, ob2 = FlatAdamW(a, **kw), torch.optim.AdamW([b], capturable=True, **kw)
    oa2.load_state_dict(sb); ob2.load_state_dict(sa)
    for g in gs[2:]:
        a.grad, b.grad = g.clone(), g.clone()
        oa2.step(); ob2.step()
    assert float(oa2.step_t) == 4.0
    assert torch.allclose(a.data, b.data, rtol=2e-6, atol=2e-7), (a.data - b.data).abs().max()


def test_flat_adamw_parameter_groups_match_torch_adamw_groups(ops):
    """optim.FlatAdamW(paramwise_cfg = the reference configs' custom_keys: decay_mult = 0 for 'norm' parameters,
    configs/swin/mask_rcnn_swin_tiny_..._1x_coco.py:64-67) -- still ONE launch over the flat buffer, the groups in a byte map --
    against torch.optim.AdamW with the per-parameter groups mmcv's optimizer constructor would build, 4 steps on the same
    gradients: every parameter to f32 rounding; plus an lr_mult group to cover the second multiplier."""
    from panoswintransformerobjectdetection_amd import SimplePanoSwinTransformer
    from panoswintransformerobjectdetection_amd.dp import GradReducer
    from panoswintransformerobjectdetection_amd.optim import REFERENCE_PARAMWISE_CFG, FlatAdamW, paramwise_groups
    cfg = dict(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], ape=True, drop_path_rate=0.0)
    pw = dict(custom_keys=dict(REFERENCE_PARAMWISE_CFG["custom_keys"], **{"abs_encoder": dict(lr_mult=0.1, decay_mult=0.5)}))
    torch.manual_seed(5)
    m = SimplePanoSwinTransformer(**cfg, compute_dtype=torch.bfloat16)
    m.init_weights(None)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(torch.randn_like(p) * 0.5)                 # norm gains / biases away from their 1 / 0 initial values
    m = m.to(DEV)
    named = list(m.named_parameters())
    ref_params = {k: torch.nn.Parameter(p.detach().clone()) for k, p in named}
    groups = {}
    for name, lr_mult, decay_mult in paramwise_groups(named, pw, "backbone"):
        groups.setdefault((lr_mult, decay_mult), []).append(ref_params[name])
    assert set(groups) == {(1.0, 1.0), (1.0, 0.0), (0.1, 0.5)}
    assert len(groups[(1.0, 0.0)]) == sum("norm" in k for k, _ in named)
    kw = dict(lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    o_ref = torch.optim.AdamW([dict(params=ps, lr=kw["lr"] * a, weight_decay=kw["weight_decay"] * b) for (a, b), ps in groups.items()],
                              betas=kw["betas"], eps=kw["eps"])
    red = GradReducer(m, pack=True)
    flat = red.flatten_parameters(m, torch.bfloat16)
    o_mine = FlatAdamW(flat, model=m, paramwise_cfg=pw, **kw)
    assert o_mine.group_of is not None and len(o_mine.group_mults) == 3
    for step in range(4):
        red.flat.copy_(torch.randn_like(red.flat) * (0.1 + step))
        for k, p in named:
            ref_params[k].grad = p._grad_slot.view_as(p).clone()          # this parameter's slice of the flat gradient buffer
        o_ref.step()
        o_mine.step()
        for k, p in named:
            assert torch.allclose(p.data, ref_params[k].data, rtol=2e-6, atol=2e-7), (step, k, (p.data - ref_params[k].data).abs().max())
    # the shadow follows the grouped update too
    assert torch.equal(m.__dict__["_flat_pair"][1], flat.data.to(torch.bfloat16))
