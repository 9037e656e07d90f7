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
