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
    _lib.call("pswin_stem_conv1_stats", x4, _ptr(x4), _ptr(w1p), B, H, W, int(want_xx), _ptr(sums), _ptr(ws))
    return sums


def conv2_fwd(x4, w1p, scale1, shift1, w2p, ws, want_stats=True):
    B, H, W, _ = x4.shape
    y2 = torch.empty(B, H, W, C2, device=x4.device, dtype=torch.bfloat16)
    sums2 = torch.empty(2 * C2, device=x4.device, dtype=torch.float32) if want_stats else None
    _lib.call("pswin_stem_conv2_fwd", x4, _ptr(x4), _ptr(w1p), _ptr(scale1), _ptr(shift1), _ptr(w2p), B, H, W, _ptr(y2),
              _ptr(sums2), _ptr(ws))
    return y2, sums2


def conv3_fwd(y2, scale2, shift2, w3p, bias3):
    B, H, W, _ = y2.shape
    tok = torch.empty(B * (H // 4) * (W // 4), C3, device=y2.device, dtype=torch.bfloat16)
    _lib.call("pswin_stem_conv3_fwd", y2, _ptr(y2), _ptr(scale2), _ptr(shift2), _ptr(w3p), _ptr(bias3), B, H, W, _ptr(tok))
    return tok


def conv3_bwd_stats(dtok, y2, prm, w3t, ws):
    B, H, W, _ = y2.shape
    sums = torch.empty(2 * C2, device=y2.device, dtype=torch.float32)
    _lib.call("pswin_stem_conv3_bwd_stats", y2, _ptr(dtok), _ptr(y2), _ptr(prm), _ptr(w3t), B, H, W, _ptr(sums), _ptr(ws))
    return sums


def conv3_bwd_data(dtok, y2, prm, w3t):
    B, H, W, _ = y2.shape
    dy2 = torch.empty_like(y2)
    _lib.call("pswin_stem_conv3_bwd_data", y2, _ptr(dtok), _ptr(y2), _ptr(prm), _ptr(w3t), B, H, W, _ptr(dy2))
    return dy2


def decode_dw3(raw):
    """accumulator tiles [ky][nh][kx][nt][mt][e][g][c] -> [96 out][64 in][4][4]; out = 16 mt + 4 g + e,
    in = 16 (2 nh + nt) + c  (wave = kx + 4 nh)"""
    t = raw.view(4, 2, 4, 2, 6, 4, 4, 16)
    return t.permute(4, 6, 5, 1, 3, 7, 0, 2).reshape(C3, C2, 4, 4)


def conv3_wgrad(dtok, y2, scale2, shift2, ws):
    B, H, W, _ = y2.shape
    raw = torch.empty(4 * 8 * 12 * 256, device=y2.device, dtype=torch.float32)
    _lib.call("pswin_stem_conv3_wgrad", y2, _ptr(dtok), _ptr(y2), _ptr(scale2), _ptr(shift2), B, H, W, _ptr(raw), _ptr(ws))
    return decode_dw3(raw)


def decode_dw2(raw):
    """accumulator tiles [mh][mi][tap][nt][e][g][c] -> [64 out][32 in][3][3]; out = 16 (2 mh + mi) + 4 g + e, in = 16 nt + c"""
    t = raw.view(2, 2, 9, 2, 4, 4, 16)
    return t.permute(0, 1, 5, 4, 3, 6, 2).reshape(C2, C1, 3, 3)


def conv2_wgrad(x4, w1p, scale1, shift1, dy2, ws):
    B, H, W, _ = x4.shape
    raw = torch.empty(2 * 36 * 256, device=x4.device, dtype=torch.float32)
    _lib.call("pswin_stem_conv2_wgrad", x4, _ptr(x4), _ptr(w1p), _ptr(scale1), _ptr(shift1), _ptr(dy2), B, H, W, _ptr(raw),
              _ptr(ws))
    return decode_dw2(raw)


def conv2_bwd(x4, w1p, prm, dy2, w2t, ws):
    B, H, W, _ = x4.shape
    out = torch.empty(2 * C1 + C1 * NSLOT, device=x4.device, dtype=torch.float32)
    _lib.call("pswin_stem_conv2_bwd", x4, _ptr(x4), _ptr(w1p), _ptr(prm), _ptr(dy2), _ptr(w2t), B, H, W, _ptr(out), _ptr(ws))
    return out[:C1], out[C1:2 * C1], out[2 * C1:].view(C1, NSLOT)


def bn_fold(sum_, sumsq, count, bn, conv_bias, training):
    """Batch statistics (training) or running statistics (eval) of a conv output whose bias was NOT applied ->
    (scale, shift, mean, rstd), all f32 [C]; updates the BatchNorm buffers like nn.BatchNorm2d in training."""
    if training:
        mean64 = sum_.double() / count
        var64 = (sumsq.double() / count - mean64 * mean64).clamp_min(0.0)
        mean, var = mean64.float(), var64.float()
        if bn.track_running_stats and bn.running_mean is not None:
            with torch.no_grad():
                bn.num_batches_tracked += 1
                m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                full_mean = mean if conv_bias is None else mean + conv_bias.detach().float()
                bn.running_mean.mul_(1 - m).add_(full_mean, alpha=m)
                bn.running_var.mul_(1 - m).add_(var * (count / max(count - 1, 1)), alpha=m)
    else:
        mean = bn.running_mean.float() if conv_bias is None else bn.running_mean.float() - conv_bias.detach().float()
        var = bn.running_var.float()
    rstd = torch.rsqrt(var + bn.eps)
    scale = bn.weight.detach().float() * rstd
    shift = bn.bias.detach().float() - mean * scale
    return scale.contiguous(), shift.contiguous(), mean.contiguous(), rstd.contiguous()


class _Stem(torch.autograd.Function):
    """tokens[B*H/4*W/4, 96] (bf16) = conv3(relu(bn2(conv2(relu(bn1(conv1(x))))))) with autograd for every parameter
    (not for x: the image).  bn1 / bn2 are the nn.BatchNorm2d modules (running statistics are updated in place)."""

    @staticmethod
    def forward(ctx, x, w1, b1, g1, be1, w2, b2, g2, be2, w3, b3, bn1, bn2, training):
        B, _, H, W = x.shape
        n = B * H * W
        ws = workspace(x)
        x4 = pack_input(x)
        w1p = pack_w1(w1)
        need_grad = any(ctx.needs_input_grad[1:11])
        if training:
            sums = conv1_stats(x4, w1p, ws, want_xx=need_grad)
            s1, q1, xx = sums[:C1], sums[C1:2 * C1], sums[2 * C1:].view(NSLOT, NSLOT)
        else:
            s1 = q1 = xx = None
        sc1, sh1, mean1, rstd1 = bn_fold(s1, q1, n, bn1, b1, training)
        y2, sums2 = conv2_fwd(x4, w1p, sc1, sh1, pack_taps(w2, False), ws, want_stats=training)
        s2, q2 = (sums2[:C2], sums2[C2:]) if training else (None, None)
        sc2, sh2, mean2, rstd2 = bn_fold(s2, q2, n, bn2, b2, training)
        tok = conv3_fwd(y2, sc2, sh2, pack_taps(w3, False), b3.detach().float().contiguous())
        ctx.training = training
        ctx.save_for_backward(x4, y2, w1p, w2, w3, sc1, sh1, mean1, rstd1, sc2, sh2, mean2, rstd2,
                              xx if xx is not None else sc1)
        return tok

    @staticmethod
    def backward(ctx, dtok):
        x4, y2, w1p, w2, w3, sc1, sh1, mean1, rstd1, sc2, sh2, mean2, rstd2, xx = ctx.saved_tensors
        training = ctx.training
        B, H, W, _ = x4.shape
        n = float(B * H * W)
        ws = torch.empty(_lib.load().pswin_stem_workspace(B, H, W), device=x4.device, dtype=torch.float32)
        dtok = dtok.to(torch.bfloat16).contiguous()
        w3t, w2t = pack_taps(w3, True), pack_taps(w2, True)
        # BN2 backward sums, then dy2
        b2c = -mean2 * rstd2
        s = conv3_bwd_stats(dtok, y2, torch.stack([sc2, sh2, rstd2, b2c]).contiguous(), w3t, ws)
        dbe2, dg2 = s[:C2].clone(), s[C2:].clone()
        if training:
            m1, m2 = dbe2 / n, dg2 / n
            P, Q = sc2 * rstd2 * m2, sc2 * (m1 + b2c * m2)
        else:
            P = Q = torch.zeros_like(sc2)
        dy2 = conv3_bwd_data(dtok, y2, torch.stack([sc2, sh2, sc2, P, Q]).contiguous(), w3t)
        dw3 = conv3_wgrad(dtok, y2, sc2, sh2, ws).clone()
        cs_ws = torch.empty(_lib.load().pswin_colsum_workspace(dtok.shape[0], C3, _lib.BF16), device=x4.device,
                            dtype=torch.float32)
        db3 = torch.empty(C3, device=x4.device, dtype=torch.float32)
        _lib.call("pswin_colsum", dtok, _ptr(dtok), _lib.BF16, dtok.shape[0], C3, _ptr(db3), _ptr(cs_ws))
        dw2 = conv2_wgrad(x4, w1p, sc1, sh1, dy2, ws).clone()
        b1c = -mean1 * rstd1
        sg, sgy, G = conv2_bwd(x4, w1p, torch.stack([sc1, sh1, rstd1, b1c]).contiguous(), dy2, w2t, ws)
        dbe1, dg1 = sg.clone(), sgy.clone()
        if training:
            # dW1 = gamma rstd (G - mean(g1) X1 - mean(g1 yhat1) Y), X1 = sum xp, Y = sum yhat1 (x) xp = rstd (W1 XX - mean X1)
            X1 = xx[ONES]
            Y = rstd1[:, None] * (w1p.float().view(C1, NSLOT) @ xx - mean1[:, None] * X1[None, :])
            dw1p = sc1[:, None] * (G - (sg / n)[:, None] * X1[None, :] - (sgy / n)[:, None] * Y)
            db1, db2 = torch.zeros_like(sg), torch.zeros_like(dbe2)      # a bias in front of a BatchNorm has no gradient
        else:
            dw1p = sc1[:, None] * G
            db1, db2 = sc1 * sg, sc2 * dbe2
        dw1 = dw1p.view(C1, 12, 4)[:, :9, :3].permute(0, 2, 1).reshape(C1, 3, 3, 3).contiguous()
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
