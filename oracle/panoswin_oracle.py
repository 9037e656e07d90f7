"""CPU oracle: a plain-PyTorch restatement of the PanoSwin windowed-attention backbone.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module -- as the checker, never as the thing measured or shipped.  The
product package ``panoswintransformerobjectdetection_amd`` never imports it and has no CPU fallback.

Pinning: this restatement is checked (tests/test_oracle_vs_reference.py, runs where
``/root/reference`` exists) against the live reference imported through ``oracle/ref_loader.py``, and
(everywhere, including the GPU box) against the fixtures in ``tests/golden/`` that
``oracle/gen_golden.py`` captured from that live reference.  The reference's own known answers are
re-checked too (HOT:105-115 relative-position index, HOT:162-171 uv grid, the WindowTransition
round-trip of HOT:1276-1283, great_circle.py:108-118 city distances).

What it follows (paths relative to /root/reference, HOT =
mmdet/models/backbones/simple_panoswin_transformer.py): HOT:44-983 and HOT:990-1237,
lzx/models/great_circle.py:71-86, lzx/pano_rotate.py:16-95,169-187, and timm's DropPath.

It is written as a *different program* from the reference on purpose: uv coordinates do not ride as two
extra channels, and every roll/flip/cat/pad/partition chain is replaced by one closed-form index map
(SURVEY.md appendix A1/A2), so agreement with the reference also validates those closed forms, which
are exactly what the HIP kernels implement.

Decisions on reference quirks (SURVEY.md D-list): D4 encoder only evaluated when ``ape``; D5 alpha/beta
are independent parameters; D6 ``train()`` returns self; D7 the pitch shortcut is LN(x); D8
``frozen_stages`` accepted and ignored; D10 the odd-width flag is derived, not stored; D12 dropout
acts on features only (all reference configs use drop_rate = 0); D16 the legacy first-size-3-dim ``cross``
is reproduced.
"""
import math
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint as checkpoint

WS_DEFAULT = 7


# ------------------------------------------------------------------------------------------------
# integer geometry (bit-exact rows of SURVEY section 8a: 1, 2, 3, 10, 12, 13)
# ------------------------------------------------------------------------------------------------
def relative_position_index(ws=WS_DEFAULT):
    """HOT:95-129.  idx[i, j] = (hi - hj + ws-1) * (2 ws - 1) + (wi - wj + ws-1), i = hi * ws + wi."""
    t = torch.arange(ws * ws)
    hi, wi = t // ws, t % ws
    dh = hi[:, None] - hi[None, :] + (ws - 1)
    dw = wi[:, None] - wi[None, :] + (ws - 1)
    return dh * (2 * ws - 1) + dw


def _ceil_to(v, m):
    return (v + m - 1) // m * m


def pano_window_map(H, W, shift, ws=WS_DEFAULT):
    """Closed form of WindowTransition.forward (pano) o pad_x o window_partition.

    HOT:393-406 (roll W by +s, ew2ns, roll H by +s), HOT:486-491 (zero pad to multiples of ws),
    HOT:64-75 (partition).  Returns (map, Hp, Wp) where map is int64 [nW * ws*ws]: the flat source
    token h*W+w for every window slot, or -1 for a zero (padding) slot.  Window order is row-major over
    the padded north-south grid, token order row-major inside the window.
    """
    Wq = W + (W % 2)            # ew2ns right-pads one zero column for odd W (HOT:344-347)
    M = Wq // 2
    SH, SW = 2 * H, M
    Hp, Wp = _ceil_to(SH, ws), _ceil_to(SW, ws)
    Y = torch.arange(Hp)[:, None].expand(Hp, Wp)
    X = torch.arange(Wp)[None, :].expand(Hp, Wp)
    inside = (Y < SH) & (X < SW)
    y = (Y - shift) % SH
    top = y < H                  # top half of the NS layout = flipped right half of the EW layout
    h = torch.where(top, H - 1 - y, y - H)
    w1 = torch.where(top, Wq - 1 - X, X)
    inside = inside & (w1 < W)   # the padded column of an odd-width map is a zero slot
    w = (w1 - shift) % W
    src = torch.where(inside, h * W + w, torch.full_like(h, -1))
    return _grid_to_windows(src, ws), Hp, Wp


def planar_window_map(H, W, shift, ws=WS_DEFAULT):
    """Closed form of pad_x o roll(-s, -s) o window_partition (HOT:522-523, 392, 64-75)."""
    Hp, Wp = _ceil_to(H, ws), _ceil_to(W, ws)
    Y = torch.arange(Hp)[:, None].expand(Hp, Wp)
    X = torch.arange(Wp)[None, :].expand(Hp, Wp)
    sy, sx = (Y + shift) % Hp, (X + shift) % Wp
    inside = (sy < H) & (sx < W)
    src = torch.where(inside, sy * W + sx, torch.full_like(sy, -1))
    return _grid_to_windows(src, ws), Hp, Wp


def _grid_to_windows(grid, ws):
    Hp, Wp = grid.shape
    g = grid.reshape(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3)
    return g.reshape(-1).contiguous()


def invert_window_map(wmap, n_tokens):
    """Slot index for every source token (every real token occupies exactly one slot)."""
    inv = torch.full((n_tokens,), -1, dtype=torch.int64)
    valid = wmap >= 0
    inv[wmap[valid]] = torch.nonzero(valid).squeeze(1)
    assert bool((inv >= 0).all()), "window map does not cover every token"
    return inv


def planar_attention_mask(H, W, shift, ws=WS_DEFAULT):
    """HOT:664-688.  Region ids over the padded grid, 0 / -100.0 (not -inf) per window pair."""
    Hp, Wp = _ceil_to(H, ws), _ceil_to(W, ws)

    def reg(v, L):
        return (v >= L - ws).long() + (v >= L - shift).long()

    rid = 3 * reg(torch.arange(Hp), Hp)[:, None] + reg(torch.arange(Wp), Wp)[None, :]
    rid = _grid_to_windows(rid, ws).reshape(-1, ws * ws)
    diff = rid[:, None, :] - rid[:, :, None]
    return torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0))


def patch_merge_map(H, W):
    """HOT:563-572.  int64 [H2*W2, 4]: source token of the 4 channel blocks, -1 = zero pad."""
    H2, W2 = (H + 1) // 2, (W + 1) // 2
    i = torch.arange(H2)[:, None].expand(H2, W2)
    j = torch.arange(W2)[None, :].expand(H2, W2)
    out = []
    for dy, dx in ((0, 0), (1, 0), (0, 1), (1, 1)):
        yy, xx = 2 * i + dy, 2 * j + dx
        ok = (yy < H) & (xx < W)
        out.append(torch.where(ok, yy * W + xx, torch.full_like(yy, -1)).reshape(-1))
    return torch.stack(out, 1)


# ------------------------------------------------------------------------------------------------
# float geometry
# ------------------------------------------------------------------------------------------------
def uv_grid(H, W, device="cpu"):
    """HOT:153-189.  [H, W, 2] = (u, v); same fp32 op order: int * gap, - pi (u) / - pi/2 (v), + gap/2."""
    assert H <= W, "make_uv_hw2 slices arange(W)[:H] (HOT:174-175)"
    gap = math.pi / H
    xs = torch.arange(W, device=device)
    ys = xs[:H]
    yy, xx = torch.meshgrid(ys, xs, indexing="ij")
    uv = torch.stack([xx, yy], -1) * gap
    uv[..., 1] -= math.pi * 0.5
    uv[..., 0] -= math.pi
    uv += 0.5 * gap
    return uv


def haversine(uv1, uv2):
    """lzx/models/great_circle.py:71-86.  uv1 [..., N, 2], uv2 [..., M, 2] -> [..., N, M]."""
    u1, v1 = uv1[..., 0][..., :, None], uv1[..., 1][..., :, None]
    u2, v2 = uv2[..., 0][..., None, :], uv2[..., 1][..., None, :]
    a = torch.sin(0.5 * torch.abs(v2 - v1)) ** 2 + torch.cos(v2) * torch.cos(v1) * torch.sin(0.5 * (u2 - u1)) ** 2
    return torch.arcsin(a ** 0.5) * 2


def abs_position_features(uv_hw2):
    """HOT:926-932: [sin u sin v, cos u sin v, cos v, u, v]."""
    u, v = uv_hw2[..., 0], uv_hw2[..., 1]
    xyz = torch.stack([torch.sin(u) * torch.sin(v), torch.cos(u) * torch.sin(v), torch.cos(v)], -1)
    return torch.cat([xyz, uv_hw2], -1)


def _sph_to_xyz(uv):
    """lzx/pano_rotate.py:16-27."""
    a = torch.sin(uv[:, 1] + math.pi * 0.5)
    return torch.stack([torch.sin(uv[:, 0]) * a, torch.cos(uv[:, 0]) * a,
                        torch.cos(uv[:, 1] + math.pi * 0.5)], -1)


def _cross_legacy(a, b):
    """torch.cross WITHOUT ``dim`` as the reference calls it (lzx/pano_rotate.py:43,46): the product is
    taken along the FIRST dimension of size 3.  Differs from dim=-1 only for a [3, 3] operand, i.e. when
    exactly two points are rotated (quirk D16; e.g. the two window centres of a 7 x 14 map).  Kept for
    bit-parity with the reference."""
    dim = [i for i, s in enumerate(a.shape) if s == 3][0]
    return torch.cross(a, b, dim=dim)


def rotate_to_pole(np_uv, s_uv, eps=1e-15):
    """lzx/pano_rotate.py:30-55 and 66-95 (forward direction)."""
    if torch.abs(np_uv[1] + math.pi * 0.5) < eps:
        return s_uv
    s_uv = torch.cat([s_uv, torch.tensor([[0.0, -0.5 * math.pi]], device=s_uv.device)], 0)
    n = _sph_to_xyz(np_uv[None, :])
    p = _sph_to_xyz(s_uv)
    dist = torch.norm(n - p, dim=1, p=2)
    v_new = 2 * torch.asin(dist / 2) - 0.5 * math.pi
    dirs = F.normalize(_cross_legacy(p, n.repeat(p.shape[0], 1)), p=2, dim=-1)
    xdir = dirs[-1]
    ydir = _cross_legacy(xdir[None], n)[0]
    u_new = torch.arccos(torch.clip((xdir[None] * dirs).sum(-1), min=-1 + eps, max=1 - eps))
    u_new = torch.where((ydir[None] * dirs).sum(-1) < 0, -u_new, u_new)
    return torch.stack([u_new[:-1], v_new[:-1]], 1)


def pitch_image_grid(Hp, Wp, np_uv):
    """Sampling grid of lzx/pano_rotate.py:169-187 for an Hp x Wp map: [1, Hp, Wp, 2] (x, y)."""
    mv, mu = torch.meshgrid(torch.arange(Hp) / Hp - 0.5, torch.arange(Wp) / Hp - 1, indexing="ij")
    s_uv = (torch.stack([mu, mv], -1) * math.pi).reshape(-1, 2)
    r = rotate_to_pole(np_uv, s_uv)
    e = 5e-4
    gx = torch.clip(r[:, 0] / math.pi, min=e - 1, max=1 - e)
    gy = torch.clip(r[:, 1] / math.pi * 2, min=e - 1, max=1 - e)
    return torch.stack([gx, gy], -1).reshape(1, Hp, Wp, 2)


def pitch_window_grid(Hp, Wp, ws, np_uv, pad_r, pad_b):
    """Second sampling grid of HOT:1040-1089: [1, nWin, ws*ws, 2]; token (p, q) of window (i, j)."""
    nH, nW = Hp // ws, Wp // ws
    us = ((torch.arange(nW) * 1.0 + 0.5) / nW * 2.0 * (1.0 - pad_r / Wp) - 1.0) * math.pi
    vs = ((torch.arange(nH) * 1.0 + 0.5) / nH * (1.0 - pad_b / Hp) - 0.5) * math.pi
    vm, um = torch.meshgrid(vs, us, indexing="ij")
    centres = rotate_to_pole(np_uv, torch.stack([um, vm], -1).reshape(-1, 2)).reshape(nH, nW, 2)
    centres = centres / math.pi
    centres = torch.stack([centres[..., 0], -centres[..., 1]], -1).flip(0)
    centres = torch.stack([centres[..., 0], centres[..., 1] * 2], -1)
    a = (torch.arange(ws) + 0.5 - 0.5 * ws) / Hp
    ax, ay = torch.meshgrid(a, a, indexing="ij")       # x offset follows the FIRST token axis
    off = torch.stack([ax, ay], -1) * 2
    off = torch.stack([off[..., 0] * 0.5, off[..., 1]], -1)
    g = (centres[:, :, None, None, :] + off[None, None]).reshape(nH * nW, ws * ws, 2)
    g = torch.where(g <= -1.0, g + 2.0, g)
    g = torch.where(g >= 1.0, g - 2.0, g)
    return g[None]


def pitch_rotate_windows(x_bchw, ws, np_uv, pad_r, pad_b):
    """HOT:1025-1097: rotate the map, re-sample one ws x ws window around every rotated centre.

    Returns [B, C, nWin, ws*ws] (window-major), i.e. already partitioned.
    """
    B, C, Hp, Wp = x_bchw.shape
    g1 = pitch_image_grid(Hp, Wp, np_uv).to(x_bchw.device)
    rot = torch.cat([F.grid_sample(x_bchw[b:b + 1], g1, mode="bilinear", padding_mode="border",
                                   align_corners=False) for b in range(B)], 0)
    g2 = pitch_window_grid(Hp, Wp, ws, np_uv, pad_r, pad_b).to(x_bchw.device)
    return F.grid_sample(rot, g2.repeat(B, 1, 1, 1), padding_mode="border", align_corners=False)


# ------------------------------------------------------------------------------------------------
# building blocks
# ------------------------------------------------------------------------------------------------
def gather_windows(x_bsc, wmap):
    """[B, S, C] -> [B * nW, ws*ws, C] through a window map (zeros in the -1 slots)."""
    B, S, C = x_bsc.shape
    idx = wmap.clamp(min=0).to(x_bsc.device)
    win = x_bsc[:, idx, :] * (wmap >= 0).to(x_bsc.dtype).to(x_bsc.device)[None, :, None]
    return win.reshape(B, -1, C)


def scatter_windows(win, inv):
    """Inverse of gather_windows for the real tokens: [B, nSlots, C] -> [B, S, C]."""
    return win[:, inv.to(win.device), :]


def drop_path(x, p, training):
    """timm.models.layers.DropPath (per sample)."""
    if p == 0.0 or not training:
        return x
    keep = 1 - p
    rnd = keep + torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), dtype=x.dtype, device=x.device)
    rnd.floor_()
    return x.div(keep) * rnd


class Mlp(nn.Module):
    """HOT:44-61."""

    def __init__(self, dim, hidden, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)
        self.drop = drop

    def forward(self, x):
        x = F.dropout(F.gelu(self.fc1(x)), self.drop, self.training)
        return F.dropout(self.fc2(x), self.drop, self.training)


def _make_tables(ws, heads):
    a = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
    b = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
    nn.init.trunc_normal_(a, std=0.02, a=-2.0, b=2.0)
    with torch.no_grad():
        b.copy_(a)               # D5: the reference's two tables start from identical values
    return a, b


def window_attention_core(q, k, v, scale, bias_neOO, mask, heads, attn_drop, training):
    """HOT:290-308.  q, k, v: [n, O, C]; bias [1|n, e, O, O]; mask None | [nW, O, O] | [B, nW, O, O]."""
    n, O, C = q.shape
    d = C // heads
    q = q.reshape(n, O, heads, d).permute(0, 2, 1, 3) * scale
    k = k.reshape(n, O, heads, d).permute(0, 2, 1, 3)
    v = v.reshape(n, O, heads, d).permute(0, 2, 1, 3)
    attn = q @ k.transpose(-2, -1) + bias_neOO
    if mask is not None:
        if mask.dim() == 3:
            mask = mask.unsqueeze(0)
        nW = mask.shape[1]
        attn = (attn.view(n // nW, nW, heads, O, O) + mask.unsqueeze(2)).view(-1, heads, O, O)
    attn = F.dropout(torch.softmax(attn, dim=-1), attn_drop, training)
    return (attn @ v).transpose(1, 2).reshape(n, O, C)


class WindowAttention(nn.Module):
    """HOT:211-323 (BasicWindowAttention + WindowAttention)."""

    def __init__(self, dim, ws, heads, qkv_bias=True, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.dim, self.ws, self.heads = dim, ws, heads
        self.scale = qk_scale or (dim // heads) ** -0.5
        self.register_buffer("relative_position_index_OO", relative_position_index(ws))
        self.proj = nn.Linear(dim, dim)
        self.sphere_position_alpha_table_Te, self.sphere_position_beta_table_Te = _make_tables(ws, heads)
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop, self.proj_drop = attn_drop, proj_drop

    def bias(self, dist_nOO, pano_mode):
        """HOT:241-272: d * alpha[idx] + beta[idx] (pano) or beta[idx] (planar) -> [n|1, e, O, O]."""
        O = self.ws * self.ws
        idx = self.relative_position_index_OO.reshape(-1)
        beta = self.sphere_position_beta_table_Te[idx].reshape(O, O, -1)
        if pano_mode:
            alpha = self.sphere_position_alpha_table_Te[idx].reshape(O, O, -1)
            b = dist_nOO[..., None] * alpha[None] + beta
        else:
            b = beta[None]
        return b.permute(0, 3, 1, 2)

    def forward(self, x_nOc, uv_nO2, mask, pano_mode):
        assert x_nOc.shape[-1] % self.heads == 0
        n, O, C = x_nOc.shape
        qkv = self.qkv(x_nOc).reshape(n, O, 3, C)
        dist = haversine(uv_nO2, uv_nO2) if pano_mode else None
        out = window_attention_core(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], self.scale,
                                    self.bias(dist, pano_mode), mask, self.heads, self.attn_drop,
                                    self.training)
        return F.dropout(self.proj(out), self.proj_drop, self.training)


class PanoSwinBlock(nn.Module):
    """HOT:412-536."""

    def __init__(self, dim, heads, ws, shift, mlp_ratio, qkv_bias, qk_scale, drop, attn_drop, drop_path_p):
        super().__init__()
        assert 0 <= shift < ws, "shift_size must in 0-window_size"
        self.dim, self.ws, self.shift, self.drop_path_p = dim, ws, shift, drop_path_p
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, ws, heads, qkv_bias, qk_scale, attn_drop, drop)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), drop)

    def forward(self, x_bsc, uv_s2, H, W, mask, pano_mode):
        B, S, C = x_bsc.shape
        assert S == H * W, "input feature has wrong size"
        if pano_mode:
            wmap, _, _ = pano_window_map(H, W, self.shift, self.ws)
            mask = None
        else:
            wmap, _, _ = planar_window_map(H, W, self.shift, self.ws)
            mask = mask if self.shift else None
        O = self.ws * self.ws
        win = gather_windows(self.norm1(x_bsc), wmap).reshape(-1, O, C)
        uv_win = gather_windows(uv_s2[None].to(x_bsc.dtype), wmap).reshape(-1, O, 2)
        uv_win = uv_win.repeat(B, 1, 1)
        att = self.attn(win, uv_win, mask, pano_mode).reshape(B, -1, C)
        att = scatter_windows(att, invert_window_map(wmap, S))
        x = x_bsc + drop_path(att, self.drop_path_p, self.training)
        return x + drop_path(self.mlp(self.norm2(x)), self.drop_path_p, self.training)


class PitchAttentionBlock(nn.Module):
    """HOT:990-1237 (appended to a stage when its depth is odd, HOT:636-647)."""

    def __init__(self, dim, heads, ws, qkv_bias, qk_scale, attn_drop, mlp_ratio, drop, np_v=-0.0001):
        super().__init__()
        self.dim, self.ws, self.heads = dim, ws, heads
        self.scale = qk_scale or (dim // heads) ** -0.5
        self.register_buffer("relative_position_index_OO", relative_position_index(ws))
        self.proj = nn.Linear(dim, dim)
        self.sphere_position_alpha_table_Te, self.sphere_position_beta_table_Te = _make_tables(ws, heads)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), drop)
        self.norm2 = nn.LayerNorm(dim)
        self.norm1 = nn.LayerNorm(dim)
        self.q_linear = nn.Linear(dim, dim, bias=qkv_bias)
        self.k_linear = nn.Linear(dim, dim, bias=qkv_bias)
        self.v_linear = nn.Linear(dim, dim, bias=qkv_bias)
        self.register_buffer("np_uv", torch.Tensor([1.0, np_v]) * math.pi)
        self.attn_drop, self.proj_drop = attn_drop, drop

    bias = WindowAttention.bias

    def forward(self, x_bsc, uv_s2, H, W, mask, pano_mode):
        B, S, C = x_bsc.shape
        assert S == H * W, "input feature has wrong size"
        ws, O = self.ws, self.ws * self.ws
        xn = self.norm1(x_bsc)                                  # D7: the shortcut is LN(x) as well
        Hp, Wp = _ceil_to(H, ws), _ceil_to(W, ws)
        pad_b, pad_r = Hp - H, Wp - W
        wmap, _, _ = planar_window_map(H, W, 0, ws)
        win = gather_windows(xn, wmap).reshape(-1, O, C)
        uv_win = gather_windows(uv_s2[None].to(xn.dtype), wmap).reshape(-1, O, 2).repeat(B, 1, 1)
        if pano_mode:
            img = F.pad(torch.cat([xn, uv_s2[None].expand(B, -1, -1).to(xn.dtype)], -1).view(B, H, W, C + 2),
                        (0, 0, 0, pad_r, 0, pad_b)).permute(0, 3, 1, 2)
            rot = pitch_rotate_windows(img, ws, self.np_uv.cpu(), pad_r, pad_b)   # [B, C+2, nWin, O]
            rot = rot.permute(0, 2, 3, 1).reshape(-1, O, C + 2)
            win_rot, uv_rot = rot[..., :C], rot[..., C:]
            dist = haversine(uv_win, uv_rot)
        else:
            win_rot, dist = win, None
        out = window_attention_core(self.q_linear(win), self.k_linear(win_rot), self.v_linear(win), self.scale,
                                    self.bias(dist, pano_mode), None, self.heads, self.attn_drop, self.training)
        out = F.dropout(self.proj(out), self.proj_drop, self.training).reshape(B, -1, C)
        out = scatter_windows(out, invert_window_map(wmap, S))
        x = xn + out                                            # DropPath(0.0) == Identity (HOT:1016)
        return x + self.mlp(self.norm2(x))


class PatchMerging(nn.Module):
    """HOT:539-576."""

    def __init__(self, dim):
        super().__init__()
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def forward(self, x_bsc, H, W):
        B, S, C = x_bsc.shape
        assert S == H * W, "input feature has wrong size"
        pm = patch_merge_map(H, W).to(x_bsc.device)
        g = x_bsc[:, pm.clamp(min=0).reshape(-1), :] * (pm.reshape(-1) >= 0).to(x_bsc.dtype)[None, :, None]
        return self.reduction(self.norm(g.reshape(B, -1, 4 * C)))


class BasicLayer(nn.Module):
    """HOT:579-724."""

    def __init__(self, dim, depth, heads, ws, mlp_ratio, qkv_bias, qk_scale, drop, attn_drop, dpr,
                 downsample, use_checkpoint):
        super().__init__()
        self.ws, self.shift, self.use_checkpoint = ws, ws // 2, use_checkpoint
        blocks = [PanoSwinBlock(dim, heads, ws, 0 if i % 2 == 0 else ws // 2, mlp_ratio, qkv_bias, qk_scale,
                                drop, attn_drop, dpr[i] if isinstance(dpr, list) else dpr)
                  for i in range(depth - depth % 2)]
        if depth % 2:
            blocks.append(PitchAttentionBlock(dim, heads, ws, qkv_bias, qk_scale, attn_drop, mlp_ratio, drop))
        self.blocks = nn.ModuleList(blocks)
        self.downsample = PatchMerging(dim) if downsample else None

    def forward(self, x_bsc, uv_s2, H, W, pano_mode):
        mask = None if pano_mode else planar_attention_mask(H, W, self.shift, self.ws).to(x_bsc.device)
        for blk in self.blocks:
            if self.use_checkpoint:
                x_bsc = checkpoint.checkpoint(blk, x_bsc, uv_s2, H, W, mask, pano_mode, use_reentrant=False)
            else:
                x_bsc = blk(x_bsc, uv_s2, H, W, mask, pano_mode)
        if self.downsample is None:
            return x_bsc, H, W, x_bsc, uv_s2, H, W
        Wh, Ww = (H + 1) // 2, (W + 1) // 2
        down = self.downsample(x_bsc, H, W)
        uv = uv_grid(Wh, Ww, x_bsc.device).view(-1, 2) if pano_mode else torch.zeros(Wh * Ww, 2, device=x_bsc.device)
        return x_bsc, H, W, down, uv, Wh, Ww


class PatchEmbed(nn.Module):
    """HOT:727-773."""

    def __init__(self, patch_size, in_chans, embed_dim, norm):
        super().__init__()
        self.patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        c = embed_dim // 3
        self.embed_dim = embed_dim
        self.proj = nn.Sequential(
            nn.Conv2d(in_chans, c, 3, 1, 1), nn.BatchNorm2d(c), nn.ReLU(inplace=True),
            nn.Conv2d(c, 2 * c, 3, 1, 1), nn.BatchNorm2d(2 * c), nn.ReLU(inplace=True),
            nn.Conv2d(2 * c, embed_dim, self.patch_size, self.patch_size))
        self.norm = nn.LayerNorm(embed_dim) if norm else None

    def forward(self, x):
        _, _, H, W = x.shape
        ph, pw = self.patch_size
        if W % pw:
            x = F.pad(x, (0, pw - W % pw))
        if H % ph:
            x = F.pad(x, (0, 0, 0, ph - H % ph))
        x = self.proj(x)
        if self.norm is not None:
            Wh, Ww = x.shape[2:]
            x = self.norm(x.flatten(2).transpose(1, 2)).transpose(1, 2).reshape(-1, self.embed_dim, Wh, Ww)
        return x


class SimplePanoSwinTransformerOracle(nn.Module):
    """HOT:779-983.  Same ctor kwargs, state-dict keys and outputs as the reference class."""

    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, depths=[2, 2, 7, 2], num_heads=[3, 6, 12, 24],
                 window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0.2, norm_layer=nn.LayerNorm, ape=False, patch_norm=True,
                 out_indices=(0, 1, 2, 3), frozen_stages=-1, use_checkpoint=False, pano_mode=True):
        super().__init__()
        assert norm_layer is nn.LayerNorm, "the oracle restates nn.LayerNorm only"
        self.num_layers, self.embed_dim, self.ape = len(depths), embed_dim, ape
        self.out_indices, self.frozen_stages, self.drop_rate = out_indices, frozen_stages, drop_rate
        self.patch_embed = PatchEmbed(patch_size, in_chans, embed_dim, patch_norm)
        if ape:
            self.abs_encoder = nn.Linear(5, embed_dim)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(BasicLayer(int(embed_dim * 2 ** i), depths[i], num_heads[i], window_size, mlp_ratio,
                                          qkv_bias, qk_scale, drop_rate, attn_drop_rate,
                                          dpr[sum(depths[:i]):sum(depths[:i + 1])],
                                          i < self.num_layers - 1, use_checkpoint))
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        for i in out_indices:
            self.add_module(f"norm{i}", nn.LayerNorm(self.num_features[i]))
        self.pano_mode = pano_mode

    def set_pano_mode(self, pano_mode=True):
        self.pano_mode = pano_mode

    def switch_pano_mode(self):
        self.set_pano_mode(not self.pano_mode)

    def init_weights(self, pretrained=None):
        """HOT:885-907 (checkpoint loading itself is out of scope: SURVEY section 8f-3)."""
        def _init(m):
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02, a=-2.0, b=2.0)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        if isinstance(pretrained, str) or pretrained is None:
            self.apply(_init)
            if isinstance(pretrained, str):
                sd = torch.load(pretrained, map_location="cpu")
                self.load_state_dict(sd.get("state_dict", sd.get("model", sd)), strict=False)
        else:
            raise TypeError('pretrained must be a str or None')

    def forward(self, x_bchw, pano_ratio_v=None):
        if pano_ratio_v is not None:
            warnings.warn("Parameter pano_ratio_v for is deprecated! Please set it to None!")
        if self.pano_mode and x_bchw.shape[3] != x_bchw.shape[2] * 2:
            warnings.warn("PanoSwin is configured in Pano mode, expecting channel3 == 2 * channel2, but get {} and {}, "
                          "probably cause an error".format(x_bchw.shape[3], x_bchw.shape[2]))
        x = self.patch_embed(x_bchw.float())
        B, C, Wh, Ww = x.shape
        if self.pano_mode:
            uv_hw2 = uv_grid(Wh, Ww, x.device)
            if self.ape:
                x = x + self.abs_encoder(abs_position_features(uv_hw2)[None]).permute(0, 3, 1, 2)
            uv = uv_hw2.reshape(-1, 2)
        else:
            uv = torch.zeros(Wh * Ww, 2, device=x.device)
        x = F.dropout(x.flatten(2).transpose(1, 2), self.drop_rate, self.training)
        outs = []
        for i, layer in enumerate(self.layers):
            x_out, H, W, x, uv, Wh, Ww = layer(x, uv, Wh, Ww, self.pano_mode)
            if i in self.out_indices:
                y = getattr(self, f"norm{i}")(x_out)
                outs.append(y.view(-1, H, W, self.num_features[i]).permute(0, 3, 1, 2).contiguous())
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        return self
