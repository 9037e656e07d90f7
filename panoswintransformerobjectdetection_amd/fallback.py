"""Plain-PyTorch window blocks for the options the HIP kernels are NOT specialised for ("any other option
falls back to the restatement running as plain torch ops on the GPU, never to an error"): ``window_size != 7``, ``head_dim != 32``,
``drop_rate`` / ``attn_drop_rate`` > 0.

Scope and rule.  Every configuration in the reference tree (configs/swin/*: window 7, head_dim 32, dropout 0) runs on the hand-written
kernels and NEVER reaches this file; a model that needs it says so once (a ``UserWarning`` naming the option) -- the HIP path is not
silently replaced.  There is no CPU fallback for the product: ``SimplePanoSwinTransformer.forward`` still refuses CPU inputs; the
functions below are device-agnostic torch code only so that they can be checked on the CPU against the oracle
(tests/test_fallback.py).  What is restated here is one PanoSwinTransformerBlock (HOT:412-536 with HOT:64-92, 211-323, 326-409, 664-688
and lzx/models/great_circle.py:71-86), generic in the window size; autograd provides the backward pass.  The stem, PatchMerging, the
output norms and the DP / optimizer code are the same as for the specialised configurations.  PitchAttentionModule (odd depths) has no
generic form here: it raises for these options.
"""
import math

import torch
import torch.nn.functional as F

_CACHE = {}


def _ceil_to(v, m):
    return (v + m - 1) // m * m


def window_map(pano, H, W, shift, ws, device):
    """(map long [nW * ws * ws] with S = H * W for a zero slot, inverse long [S], nW): WindowTransition + pad + window_partition as one
    index map, SURVEY.md appendix A1 (pano: roll W, ew2ns, roll H; odd W gets one zero column) / A2 (planar: pad, then roll)."""
    key = ("map", bool(pano), H, W, shift, ws, str(device))
    if key in _CACHE:
        return _CACHE[key]
    S = H * W
    if pano:
        Wq = W + (W % 2)
        SH, SW = 2 * H, Wq // 2
        Hp, Wp = _ceil_to(SH, ws), _ceil_to(SW, ws)
        Y = torch.arange(Hp, device=device)[:, None].expand(Hp, Wp)
        X = torch.arange(Wp, device=device)[None, :].expand(Hp, Wp)
        y = (Y - shift) % SH
        top = y < H
        h = torch.where(top, H - 1 - y, y - H)
        w1 = torch.where(top, Wq - 1 - X, X)
        inside = (Y < SH) & (X < SW) & (w1 < W)
        src = torch.where(inside, h * W + (w1 - shift) % W, torch.full_like(h, S))
    else:
        Hp, Wp = _ceil_to(H, ws), _ceil_to(W, ws)
        Y = torch.arange(Hp, device=device)[:, None].expand(Hp, Wp)
        X = torch.arange(Wp, device=device)[None, :].expand(Hp, Wp)
        sy, sx = (Y + shift) % Hp, (X + shift) % Wp
        src = torch.where((sy < H) & (sx < W), sy * W + sx, torch.full_like(sy, S))
    wmap = src.reshape(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1).contiguous()
    inv = torch.empty(S + 1, dtype=torch.long, device=device)
    inv[wmap] = torch.arange(wmap.numel(), device=device)
    _CACHE[key] = (wmap, inv[:S].contiguous(), wmap.numel() // (ws * ws))
    return _CACHE[key]


def relative_position_index(ws, device):
    """HOT:95-129: idx[i, j] = (h_i - h_j + ws - 1) (2 ws - 1) + (w_i - w_j + ws - 1)."""
    key = ("rpi", ws, str(device))
    if key not in _CACHE:
        t = torch.arange(ws * ws, device=device)
        hi, wi = t // ws, t % ws
        _CACHE[key] = ((hi[:, None] - hi[None, :] + ws - 1) * (2 * ws - 1) + (wi[:, None] - wi[None, :] + ws - 1)).reshape(-1)
    return _CACHE[key]


def planar_mask(H, W, shift, ws, device):
    """HOT:664-688: 0 / -100.0 per token pair of a window from the nine shifted-window regions of the padded grid."""
    key = ("mask", H, W, shift, ws, str(device))
    if key not in _CACHE:
        Hp, Wp = _ceil_to(H, ws), _ceil_to(W, ws)

        def reg(v, L):
            return (v >= L - ws).long() + (v >= L - shift).long()

        rid = 3 * reg(torch.arange(Hp, device=device), Hp)[:, None] + reg(torch.arange(Wp, device=device), Wp)[None, :]
        rid = rid.reshape(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
        _CACHE[key] = (rid[:, None, :] != rid[:, :, None]).float() * -100.0
    return _CACHE[key]


def window_distance(H, W, shift, ws, device):
    """[nW, O, O] great-circle distances between the tokens of a pano window (haversine22, great_circle.py:71-86) on the pixel-centre
    uv grid of HOT:153-189; padding slots sit at uv = (0, 0)."""
    key = ("dist", H, W, shift, ws, str(device))
    if key not in _CACHE:
        wmap, _, nW = window_map(True, H, W, shift, ws, device)
        gap = math.pi / H
        xs = torch.arange(W, device=device, dtype=torch.float32) * gap - math.pi + 0.5 * gap
        ys = torch.arange(H, device=device, dtype=torch.float32) * gap - math.pi / 2 + 0.5 * gap
        uv = torch.stack([xs[None, :].expand(H, W), ys[:, None].expand(H, W)], -1).reshape(-1, 2)
        uv = torch.cat([uv, uv.new_zeros(1, 2)], 0)[wmap].view(nW, ws * ws, 2)
        u1, v1, u2, v2 = uv[:, :, None, 0], uv[:, :, None, 1], uv[:, None, :, 0], uv[:, None, :, 1]
        a = torch.sin((v2 - v1).abs() / 2) ** 2 + torch.cos(v2) * torch.cos(v1) * torch.sin((u2 - u1) / 2) ** 2
        _CACHE[key] = 2 * torch.asin(torch.sqrt(a))
    return _CACHE[key]


def _drop_path(x, p, training):
    if p == 0.0 or not training:
        return x
    keep = 1.0 - p
    return x * (torch.floor(keep + torch.rand(x.shape[0], 1, 1, dtype=x.dtype, device=x.device)) / keep)


def block_forward(blk, x, H, W, attn_drop=0.0, drop=0.0):
    """One PanoSwinTransformerBlock (`blk`: the product's parameter holder) on x [B, H*W, C] in fp32 torch ops."""
    B, S, C = x.shape
    assert S == H * W, "input feature has wrong size"
    a, ws, dev = blk.attn, blk.window_size, x.device
    O, heads = ws * ws, a.num_heads
    pano = bool(blk.pano_mode)
    wmap, inv, nW = window_map(pano, H, W, blk.shift_size, ws, dev)
    xn = F.layer_norm(x, (C,), blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
    win = torch.cat([xn, xn.new_zeros(B, 1, C)], 1)[:, wmap].reshape(B * nW, O, C)               # zero rows in the padding slots
    qkv = F.linear(win, a.qkv.weight, a.qkv.bias).reshape(B * nW, O, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    idx = relative_position_index(ws, dev)
    bias = a.sphere_position_beta_table_Te[idx].reshape(O, O, heads).permute(2, 0, 1)[None]       # [1, e, O, O]
    if pano:                                                                                         # HOT:250-259
        alpha = a.sphere_position_alpha_table_Te[idx].reshape(O, O, heads).permute(2, 0, 1)
        bias = window_distance(H, W, blk.shift_size, ws, dev)[:, None] * alpha[None] + bias          # [nW, e, O, O]
    attn = (qkv[0] * a.scale) @ qkv[1].transpose(-2, -1)
    attn = attn.view(B, -1, heads, O, O) + bias[None]
    if not pano and blk.shift_size:                                                                  # HOT:474, 295-303
        attn = attn + planar_mask(H, W, blk.shift_size, ws, dev)[None, :, None]
    attn = F.dropout(torch.softmax(attn.view(B * nW, heads, O, O), -1), attn_drop, blk.training)
    out = (attn @ qkv[2]).transpose(1, 2).reshape(B, nW * O, C)
    out = F.dropout(F.linear(out, a.proj.weight, a.proj.bias), drop, blk.training)
    x = x + _drop_path(out[:, inv], blk.drop_path_p, blk.training)                                  # reverse + crop + reverse shift
    h = F.layer_norm(x, (C,), blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
    h = F.dropout(F.gelu(F.linear(h, blk.mlp.fc1.weight, blk.mlp.fc1.bias)), drop, blk.training)
    h = F.dropout(F.linear(h, blk.mlp.fc2.weight, blk.mlp.fc2.bias), drop, blk.training)
    return x + _drop_path(h, blk.drop_path_p, blk.training)
