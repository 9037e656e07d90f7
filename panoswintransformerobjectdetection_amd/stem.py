"""Fused PatchEmbed stem (HOT:742-750) on the HIP kernels of csrc/pswin_stem.hip: host side.

The kernels compute; this module only repacks the (tiny) weights, folds the BatchNorm statistics the kernels
return into per-channel scale / shift vectors, keeps nn.BatchNorm2d's running statistics, and wires the backward
pass into autograd.  bf16 operands, f32 accumulation -- used by PatchEmbed when compute_dtype is bfloat16 and
the layer has the reference's default geometry (in_chans 3, embed_dim 96, patch 4)."""
import torch

from . import _lib

C1, C2, C3 = 32, 64, 96
NSLOT = 48            # 12 tap slots x 4 channel slots of the 3x3x3 input patch (taps 9..11 and channel 3 unused by conv1)
ONES = 4 * 4 + 3      # centre tap, channel 3: the constant-one slot of the packed input


def _ptr(t):
    return None if t is None else t.data_ptr()


def pack_w1(w1):
    """[32,3,3,3] -> [32][12][4] bf16, slot = (ky*3+kx, channel)"""
    p = torch.zeros(C1, 12, 4, device=w1.device, dtype=torch.bfloat16)
    p[:, :9, :3] = w1.permute(0, 2, 3, 1).reshape(C1, 9, 3).to(torch.bfloat16)
    return p


def pack_taps(w, transpose):
    """[O,I,kh,kw] -> [kh*kw][O][I] (or [kh*kw][I][O]) bf16"""
    O, I, kh, kw = w.shape
    if transpose:
        return w.permute(2, 3, 1, 0).reshape(kh * kw, I, O).to(torch.bfloat16).contiguous()
    return w.permute(2, 3, 0, 1).reshape(kh * kw, O, I).to(torch.bfloat16).contiguous()


def workspace(x):
    B, _, H, W = x.shape
    n = _lib.load().pswin_stem_workspace(B, H, W)
    return torch.empty(n, device=x.device, dtype=torch.float32)


def pack_input(x):
    B, _, H, W = x.shape
    x = x.float().contiguous()
    x4 = torch.empty(B, H, W, 4, device=x.device, dtype=torch.bfloat16)
    _lib.call("pswin_stem_pack_input", x, _ptr(x), B, H, W, _ptr(x4))
    return x4


def conv1_stats(x4, w1p, ws, want_xx=True):
    B, H, W, _ = x4.shape
    sums = torch.empty(2 * C1 + NSLOT * NSLOT, device=x4.device, dtype=torch.float32)
    _lib.call("pswin_stem_conv1_stats", x4, _ptr(x4), _ptr(w1p), B, H, W, int(want_xx), _ptr(sums), _ptr(ws), algo_bytes=B * H * W * 8)
    return sums


def conv2_fwd(x4, w1p, scale1, shift1, w2p, ws, want_stats=True):
    B, H, W, _ = x4.shape
    y2 = torch.empty(B, H, W, C2, device=x4.device, dtype=torch.bfloat16)
    sums2 = torch.empty(2 * C2, device=x4.device, dtype=torch.float32) if want_stats else None
    _lib.call("pswin_stem_conv2_fwd", x4, _ptr(x4), _ptr(w1p), _ptr(scale1), _ptr(shift1), _ptr(w2p), B, H, W, _ptr(y2),
              _ptr(sums2), _ptr(ws), algo_bytes=B * H * W * 136)
    return y2, sums2


def conv3_fwd(y2, scale2, shift2, w3p, bias3):
    B, H, W, _ = y2.shape
    tok = torch.empty(B * (H // 4) * (W // 4), C3, device=y2.device, dtype=torch.bfloat16)
    _lib.call("pswin_stem_conv3_fwd", y2, _ptr(y2), _ptr(scale2), _ptr(shift2), _ptr(w3p), _ptr(bias3), B, H, W, _ptr(tok), algo_bytes=B * H * W * 128 + tok.numel() * 2)
    return tok


def conv3_bwd_stats(dtok, y2, prm, w3t, ws):
    B, H, W, _ = y2.shape
    sums = torch.empty(2 * C2, device=y2.device, dtype=torch.float32)
    _lib.call("pswin_stem_conv3_bwd_stats", y2, _ptr(dtok), _ptr(y2), _ptr(prm), _ptr(w3t), B, H, W, _ptr(sums), _ptr(ws), algo_bytes=B * H * W * 128 + dtok.numel() * 2)
    return sums


def conv3_bwd_data(dtok, y2, prm, w3t):
    B, H, W, _ = y2.shape
    dy2 = torch.empty_like(y2)
    _lib.call("pswin_stem_conv3_bwd_data", y2, _ptr(dtok), _ptr(y2), _ptr(prm), _ptr(w3t), B, H, W, _ptr(dy2), algo_bytes=B * H * W * 256 + dtok.numel() * 2)
    return dy2


def decode_dw3(raw):
    """accumulator tiles [ky][nh][kx][nt][mt][e][g][c] -> [96 out][64 in][4][4]; out = 16 mt + 4 g + e,
    in = 16 (2 nh + nt) + c  (wave = kx + 4 nh)"""
    t = raw.view(4, 2, 4, 2, 6, 4, 4, 16)
    return t.permute(4, 6, 5, 1, 3, 7, 0, 2).reshape(C3, C2, 4, 4)


_PERMS = {}


def _perm(name, decode, n, device):
    """int32 table: accumulator element -> position in the parameter layout (the inverse of decode_*), cached per device"""
    key = (name, str(device))
    if key not in _PERMS:
        src = decode(torch.arange(n, device=device, dtype=torch.float32)).reshape(-1).long()   # src[j]: accumulator index of output j
        perm = torch.empty(n, device=device, dtype=torch.int32)
        perm[src] = torch.arange(n, device=device, dtype=torch.int32)
        _PERMS[key] = perm
    return _PERMS[key]


def conv3_wgrad(dtok, y2, scale2, shift2, ws, direct=True):
    B, H, W, _ = y2.shape
    n = 4 * 8 * 12 * 256
    raw = torch.empty(n, device=y2.device, dtype=torch.float32)
    perm = _perm("dw3", decode_dw3, n, y2.device) if direct else None
    _lib.call("pswin_stem_conv3_wgrad", y2, _ptr(dtok), _ptr(y2), _ptr(scale2), _ptr(shift2), B, H, W, _ptr(perm), _ptr(raw),
              _ptr(ws), algo_bytes=B * H * W * 128 + dtok.numel() * 2)
    return raw.view(C3, C2, 4, 4) if direct else decode_dw3(raw)


def decode_dw2(raw):
    """accumulator tiles [mh][mi][tap][nt][e][g][c] -> [64 out][32 in][3][3]; out = 16 (2 mh + mi) + 4 g + e, in = 16 nt + c"""
    t = raw.view(2, 2, 9, 2, 4, 4, 16)
    return t.permute(0, 1, 5, 4, 3, 6, 2).reshape(C2, C1, 3, 3)


def conv2_wgrad(x4, w1p, scale1, shift1, dy2, ws, direct=True):
    B, H, W, _ = x4.shape
    n = 2 * 36 * 256
    raw = torch.empty(n, device=x4.device, dtype=torch.float32)
    perm = _perm("dw2", decode_dw2, n, x4.device) if direct else None
    _lib.call("pswin_stem_conv2_wgrad", x4, _ptr(x4), _ptr(w1p), _ptr(scale1), _ptr(shift1), _ptr(dy2), B, H, W, _ptr(perm),
              _ptr(raw), _ptr(ws), algo_bytes=B * H * W * 136)
    return raw.view(C2, C1, 3, 3) if direct else decode_dw2(raw)


def conv2_bwd_raw(x4, w1p, prm, dy2, w2t, ws):
    B, H, W, _ = x4.shape
    out = torch.empty(2 * C1 + C1 * NSLOT, device=x4.device, dtype=torch.float32)
    _lib.call("pswin_stem_conv2_bwd", x4, _ptr(x4), _ptr(w1p), _ptr(prm), _ptr(dy2), _ptr(w2t), B, H, W, _ptr(out), _ptr(ws), algo_bytes=B * H * W * 136)
    return out


def conv2_bwd(x4, w1p, prm, dy2, w2t, ws):
    out = conv2_bwd_raw(x4, w1p, prm, dy2, w2t, ws)
    return out[:C1], out[C1:2 * C1], out[2 * C1:].view(C1, NSLOT)


def pack_weights(w1, w2, w3):
    """the five packed bf16 operands from the fp32 parameters: one launch"""
    dev = w1.device
    bf = torch.bfloat16
    w1p = torch.empty(C1, 12, 4, device=dev, dtype=bf)
    w2p, w2t = torch.empty(9, C2, C1, device=dev, dtype=bf), torch.empty(9, C1, C2, device=dev, dtype=bf)
    w3p, w3t = torch.empty(16, C3, C2, device=dev, dtype=bf), torch.empty(16, C2, C3, device=dev, dtype=bf)
    w1c, w2c, w3c = w1.detach().float().contiguous(), w2.detach().float().contiguous(), w3.detach().float().contiguous()
    _lib.call("pswin_stem_pack_weights", w1c, _ptr(w1c), _ptr(w2c), _ptr(w3c), _ptr(w1p), _ptr(w2p), _ptr(w2t), _ptr(w3p),
              _ptr(w3t))
    return w1p, w2p, w2t, w3p, w3t


def bn_fold_prm(sums, count, bn, conv_bias, training):
    """-> prm f32 [4][C] = scale, shift, rstd, -mean rstd (one launch; updates bn's running statistics in training)"""
    C = bn.num_features
    prm = torch.empty(4, C, device=bn.weight.device, dtype=torch.float32)
    momentum = bn.momentum
    nbt = None
    if training and bn.track_running_stats:
        if momentum is None:                          # cumulative average: the host needs the count
            with torch.no_grad():
                bn.num_batches_tracked += 1
            momentum = 1.0 / float(bn.num_batches_tracked)
        else:
            nbt = bn.num_batches_tracked              # incremented by the fold kernel
    s = None if sums is None else sums[:C]
    q = None if sums is None else sums[C:2 * C]
    cb = None if conv_bias is None else conv_bias.detach()
    _lib.call("pswin_stem_bn_fold", prm, _ptr(s), _ptr(q), float(count), _ptr(bn.weight.detach()), _ptr(bn.bias.detach()),
              _ptr(cb), float(bn.eps), float(momentum or 0.0), int(training), _ptr(bn.running_mean), _ptr(bn.running_var), C,
              _ptr(prm), _ptr(nbt))
    return prm


class _Stem(torch.autograd.Function):
    """tokens[B*H/4*W/4, 96] (bf16) = conv3(relu(bn2(conv2(relu(bn1(conv1(x))))))) with autograd for every parameter
    (not for x: the image).  bn1 / bn2 are the nn.BatchNorm2d modules (running statistics are updated in place)."""

    @staticmethod
    def forward(ctx, x, w1, b1, g1, be1, w2, b2, g2, be2, w3, b3, bn1, bn2, training):
        B, _, H, W = x.shape
        n = B * H * W
        ws = workspace(x)
        x4 = pack_input(x)
        w1p, w2p, w2t, w3p, w3t = pack_weights(w1, w2, w3)
        need_grad = any(ctx.needs_input_grad[1:11])
        sums1 = conv1_stats(x4, w1p, ws, want_xx=need_grad) if training else None
        prm1 = bn_fold_prm(sums1, n, bn1, b1, training)
        y2, sums2 = conv2_fwd(x4, w1p, prm1[0], prm1[1], w2p, ws, want_stats=training)
        prm2 = bn_fold_prm(sums2, n, bn2, b2, training)
        tok = conv3_fwd(y2, prm2[0], prm2[1], w3p, b3.detach().float().contiguous())
        ctx.training = training
        ctx.save_for_backward(x4, y2, w1p, w2t, w3t, prm1, prm2, sums1 if sums1 is not None else prm1)
        return tok

    @staticmethod
    def backward(ctx, dtok):
        x4, y2, w1p, w2t, w3t, prm1, prm2, sums1 = ctx.saved_tensors
        training = ctx.training
        B, H, W, _ = x4.shape
        n = float(B * H * W)
        dev = x4.device
        ws = torch.empty(_lib.load().pswin_stem_workspace(B, H, W), device=dev, dtype=torch.float32)
        dtok = dtok.to(torch.bfloat16).contiguous()
        # BN2 backward sums (= dbeta2, dgamma2), then dy2
        s2 = conv3_bwd_stats(dtok, y2, prm2, w3t, ws)
        prm5 = torch.empty(5, C2, device=dev, dtype=torch.float32)
        _lib.call("pswin_stem_bn2_coefs", s2, _ptr(s2), _ptr(prm2), n, int(training), _ptr(prm5))
        dy2 = conv3_bwd_data(dtok, y2, prm5, w3t)
        dw3 = conv3_wgrad(dtok, y2, prm2[0], prm2[1], ws)
        cs_ws = torch.empty(_lib.load().pswin_colsum_workspace(dtok.shape[0], C3, _lib.BF16), device=dev, dtype=torch.float32)
        db3 = torch.empty(C3, device=dev, dtype=torch.float32)
        _lib.call("pswin_colsum", dtok, _ptr(dtok), _lib.BF16, dtok.shape[0], C3, _ptr(db3), _ptr(cs_ws))
        dw2 = conv2_wgrad(x4, w1p, prm1[0], prm1[1], dy2, ws)
        out5 = conv2_bwd_raw(x4, w1p, prm1, dy2, w2t, ws)
        # dW1 = gamma rstd (G - mean(g1) X1 - mean(g1 yhat1) Y), X1 = sum xp, Y = sum yhat1 (x) xp = rstd (W1 XX - mean X1)
        dw1 = torch.empty(C1, 3, 3, 3, device=dev, dtype=torch.float32)
        db1 = torch.empty(C1, device=dev, dtype=torch.float32)
        xx = sums1[2 * C1:] if training else None
        _lib.call("pswin_stem_conv1_wgrad", out5, _ptr(out5), _ptr(xx), _ptr(w1p), _ptr(prm1), n, int(training), _ptr(dw1),
                  _ptr(db1))
        dbe2, dg2 = s2[:C2], s2[C2:]
        dbe1, dg1 = out5[:C1], out5[C1:2 * C1]
        db2 = torch.zeros_like(dbe2) if training else prm2[0] * dbe2     # a bias in front of a BatchNorm has no gradient
        return None, dw1, db1, dg1, dbe1, dw2, db2, dg2, dbe2, dw3, db3, None, None, None


def stem_supported(proj, x, cd):
    """The fused kernels cover the reference's default stem: Conv(3->32,3x3) BN ReLU Conv(32->64,3x3) BN ReLU
    Conv(64->96,4x4/4) on a CUDA image that does not need a gradient, bf16 compute."""
    import torch.nn as nn
    if cd != torch.bfloat16 or not x.is_cuda or x.requires_grad or len(proj) != 7:
        return False
    c1, n1, _, c2, n2, _, c3 = proj
    ok = (isinstance(c1, nn.Conv2d) and isinstance(c2, nn.Conv2d) and isinstance(c3, nn.Conv2d)
          and isinstance(n1, nn.BatchNorm2d) and isinstance(n2, nn.BatchNorm2d))
    if not ok:
        return False
    return (tuple(c1.weight.shape) == (C1, 3, 3, 3) and tuple(c2.weight.shape) == (C2, C1, 3, 3)
            and tuple(c3.weight.shape) == (C3, C2, 4, 4) and c3.stride == (4, 4) and c1.bias is not None
            and c2.bias is not None and c3.bias is not None and n1.affine and n2.affine
            and n1.track_running_stats and n2.track_running_stats
            and x.shape[1] == 3 and x.shape[2] % 4 == 0 and x.shape[3] % 4 == 0
            and x.shape[2] * x.shape[3] * 128 < 0xFFFFFF00)


def stem_forward(proj, x, training):
    c1, n1, _, c2, n2, _, c3 = proj
    return _Stem.apply(x, c1.weight, c1.bias, n1.weight, n1.bias, c2.weight, c2.bias, n2.weight, n2.bias, c3.weight,
                       c3.bias, n1, n2, training)
