"""Host-side construction of the pitch module's static sampling tables.

PitchAttentionModule (HOT:990-1237) rotates the feature map to a new pole with F.grid_sample and then
re-samples one 7x7 window around every rotated window centre with a second F.grid_sample.  Both grids
depend only on (H, W, window padding, pole), never on the input, so they are computed once per shape on
the host and converted to 4-tap (index, weight) tables that the HIP row-interpolation kernel
(pswin_interp_rows) consumes.  The arithmetic follows lzx/pano_rotate.py:16-95,169-187 and HOT:1040-1089
step by step in fp32 so that the tables agree with what the reference feeds to grid_sample.
HOT = mmdet/models/backbones/simple_panoswin_transformer.py of the reference.
"""
import math

import torch
import torch.nn.functional as F


def _unit_vectors(uv):
    # lzx/pano_rotate.py:23-26 (x = sin u sin(v + pi/2), y = cos u sin(v + pi/2), z = cos(v + pi/2))
    s = torch.sin(uv[:, 1] + math.pi * 0.5)
    return torch.stack([torch.sin(uv[:, 0]) * s, torch.cos(uv[:, 0]) * s, torch.cos(uv[:, 1] + math.pi * 0.5)], -1)


def _cross_first_dim3(a, b):
    # the reference calls torch.cross without `dim` (lzx/pano_rotate.py:43,46): first dimension of size 3
    dim = next(i for i, s in enumerate(a.shape) if s == 3)
    return torch.cross(a, b, dim=dim)


def rotate_uv(pole_uv, pts_uv, eps=1e-15):
    """New (u, v) of points after moving the north pole to pole_uv (lzx/pano_rotate.py:30-55, 66-95)."""
    if torch.abs(pole_uv[1] + math.pi * 0.5) < eps:
        return pts_uv
    pts = torch.cat([pts_uv, torch.tensor([[0.0, -0.5 * math.pi]])], 0)     # sentinel: the old north pole
    n = _unit_vectors(pole_uv[None, :])
    p = _unit_vectors(pts)
    chord = torch.norm(n - p, dim=1, p=2)
    v_new = 2 * torch.asin(chord / 2) - 0.5 * math.pi
    dirs = F.normalize(_cross_first_dim3(p, n.repeat(p.shape[0], 1)), p=2, dim=-1)
    x_dir = dirs[-1]
    y_dir = _cross_first_dim3(x_dir[None], n)[0]
    u_new = torch.arccos(torch.clip((x_dir[None] * dirs).sum(-1), min=-1 + eps, max=1 - eps))
    u_new = torch.where((y_dir[None] * dirs).sum(-1) < 0, -u_new, u_new)
    return torch.stack([u_new[:-1], v_new[:-1]], 1)


def map_rotation_grid(Hp, Wp, pole_uv):
    """Normalised sampling grid of pano_rotate_image (lzx/pano_rotate.py:169-187): [Hp*Wp, 2] (x, y)."""
    vv, uu = torch.meshgrid(torch.arange(Hp) / Hp - 0.5, torch.arange(Wp) / Hp - 1, indexing="ij")
    pts = (torch.stack([uu, vv], -1) * math.pi).reshape(-1, 2)
    rot = rotate_uv(pole_uv, pts)
    e = 5e-4
    return torch.stack([torch.clip(rot[:, 0] / math.pi, min=e - 1, max=1 - e),
                        torch.clip(rot[:, 1] / math.pi * 2, min=e - 1, max=1 - e)], -1)


def window_resample_grid(Hp, Wp, ws, pole_uv, pad_r, pad_b):
    """Normalised grid of the per-window resampling (HOT:1040-1089): [nWin*ws*ws, 2], window-major."""
    nH, nW = Hp // ws, Wp // ws
    us = ((torch.arange(nW) * 1.0 + 0.5) / nW * 2.0 * (1.0 - pad_r / Wp) - 1.0) * math.pi
    vs = ((torch.arange(nH) * 1.0 + 0.5) / nH * (1.0 - pad_b / Hp) - 0.5) * math.pi
    vm, um = torch.meshgrid(vs, us, indexing="ij")
    ctr = rotate_uv(pole_uv, torch.stack([um, vm], -1).reshape(-1, 2)).reshape(nH, nW, 2) / math.pi
    ctr = torch.stack([ctr[..., 0], -ctr[..., 1]], -1).flip(0)
    ctr = torch.stack([ctr[..., 0], ctr[..., 1] * 2], -1)
    a = (torch.arange(ws) + 0.5 - 0.5 * ws) / Hp
    ox, oy = torch.meshgrid(a, a, indexing="ij")         # the x offset runs along the FIRST token axis (HOT:1072)
    off = torch.stack([ox, oy], -1) * 2
    off = torch.stack([off[..., 0] * 0.5, off[..., 1]], -1)
    g = (ctr[:, :, None, None, :] + off[None, None]).reshape(-1, 2)
    g = torch.where(g <= -1.0, g + 2.0, g)
    g = torch.where(g >= 1.0, g - 2.0, g)
    return g


def bilinear_taps(grid_xy, Hs, Ws):
    """(idx int32 [P, 4], wgt f32 [P, 4]) of F.grid_sample(bilinear, border, align_corners=False) on an Hs x Ws
    row-major source: taps in ATen's order nw, ne, sw, se; out-of-range taps get weight 0."""
    gx, gy = grid_xy[:, 0].float(), grid_xy[:, 1].float()
    ix = torch.clamp(((gx + 1) * Ws - 1) / 2, min=0, max=Ws - 1)
    iy = torch.clamp(((gy + 1) * Hs - 1) / 2, min=0, max=Hs - 1)
    x0, y0 = torch.floor(ix), torch.floor(iy)
    x1, y1 = x0 + 1, y0 + 1
    w = torch.stack([(x1 - ix) * (y1 - iy), (ix - x0) * (y1 - iy), (x1 - ix) * (iy - y0), (ix - x0) * (iy - y0)], -1)
    xs = torch.stack([x0, x1, x0, x1], -1).long()
    ys = torch.stack([y0, y0, y1, y1], -1).long()
    ok = (xs >= 0) & (xs < Ws) & (ys >= 0) & (ys < Hs)
    idx = torch.where(ok, ys * Ws + xs, torch.zeros_like(xs))
    w = torch.where(ok, w, torch.zeros_like(w))
    return idx.to(torch.int32).contiguous(), w.float().contiguous()


def pitch_tables(H, W, ws, pole_uv):
    """Static tables of a pitch block on an H x W token map.

    Returns dict(Hp, Wp, pad_r, pad_b, idx1, w1, idx2, w2): stage 1 reads the UNPADDED token map (taps that fall
    on zero padding get weight 0) and writes the rotated Hp x Wp map; stage 2 reads that map and writes
    nWin*49 window slots.
    """
    Hp, Wp = (H + ws - 1) // ws * ws, (W + ws - 1) // ws * ws
    pad_b, pad_r = Hp - H, Wp - W
    pole = pole_uv.detach().float().cpu()
    idx1, w1 = bilinear_taps(map_rotation_grid(Hp, Wp, pole), Hp, Wp)
    yy, xx = idx1.long() // Wp, idx1.long() % Wp
    real = (yy < H) & (xx < W)
    idx1 = torch.where(real, yy * W + xx, torch.zeros_like(yy)).to(torch.int32).contiguous()
    w1 = torch.where(real, w1, torch.zeros_like(w1)).contiguous()
    idx2, w2 = bilinear_taps(window_resample_grid(Hp, Wp, ws, pole, pad_r, pad_b), Hp, Wp)
    return dict(Hp=Hp, Wp=Wp, pad_r=pad_r, pad_b=pad_b, idx1=idx1, w1=w1, idx2=idx2, w2=w2)
