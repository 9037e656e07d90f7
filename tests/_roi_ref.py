"""Plain PyTorch statement of RoIAlign over an FPN pyramid (CHECKER for csrc/pswin_roi.hip; also what the CPU tests of the head
stand-ins plug into detector.MiniMaskRCNN).  The operator is mmcv.ops.RoIAlign as published (aligned = True, sampling_ratio = 0 ->
adaptive ceil(roi / output) grid, out-of-range samples contribute 0, coordinates clamped to the map, bilinear read, mean over the
grid) applied per RoI on the level SingleRoIExtractor.map_roi_levels assigns.  mmcv.ops is not in the reference tree: this is a
restatement of the definition, NOT a reference output (parity unpinned)."""
import math

import torch


def map_roi_levels(rois5, num_levels, finest_scale=56):
    scale = torch.sqrt((rois5[:, 3] - rois5[:, 1]) * (rois5[:, 4] - rois5[:, 2]))
    return torch.floor(torch.log2(scale / finest_scale + 1e-6)).clamp(min=0, max=num_levels - 1).long()


def _bilinear(f, y, x):
    """f [C, H, W]; y [n], x [m] sample coordinates -> [C, n, m] with RoIAlign's border rules."""
    C, H, W = f.shape
    vy, vx = ~((y < -1.0) | (y > H)), ~((x < -1.0) | (x > W))
    y, x = y.clamp(min=0), x.clamp(min=0)
    yl, xl = y.floor().long(), x.floor().long()
    top, right = yl >= H - 1, xl >= W - 1
    yl, xl = torch.where(top, torch.full_like(yl, H - 1), yl), torch.where(right, torch.full_like(xl, W - 1), xl)
    yh, xh = torch.where(top, yl, yl + 1), torch.where(right, xl, xl + 1)
    y, x = torch.where(top, yl.to(y.dtype), y), torch.where(right, xl.to(x.dtype), x)
    ly, lx = y - yl, x - xl
    hy, hx = 1 - ly, 1 - lx
    out = (f[:, yl][:, :, xl] * (hy[:, None] * hx[None]) + f[:, yl][:, :, xh] * (hy[:, None] * lx[None]) +
           f[:, yh][:, :, xl] * (ly[:, None] * hx[None]) + f[:, yh][:, :, xh] * (ly[:, None] * lx[None]))
    return out * (vy[:, None] & vx[None]).to(out.dtype)


def roi_align_fpn(feats, strides, rois5, P, sampling_ratio=0, aligned=True, finest_scale=56):
    """feats: NCHW maps; rois5 [R, 5] (batch index, x1, y1, x2, y2) -> [R, C, P, P] in feats[0].dtype, arithmetic in f32 (differentiable)."""
    lvl = map_roi_levels(rois5, len(feats), finest_scale)
    outs = []
    for r in range(rois5.shape[0]):
        l = int(lvl[r])
        f = feats[l][int(rois5[r, 0])].float()
        s, off = 1.0 / strides[l], (0.5 if aligned else 0.0)
        x1, y1, x2, y2 = [float(v) * s - off for v in rois5[r, 1:]]
        rw, rh = x2 - x1, y2 - y1
        if not aligned:
            rw, rh = max(rw, 1.0), max(rh, 1.0)
        gh = sampling_ratio if sampling_ratio > 0 else max(int(math.ceil(rh / P)), 0)
        gw = sampling_ratio if sampling_ratio > 0 else max(int(math.ceil(rw / P)), 0)
        if gh * gw == 0:
            outs.append(f.new_zeros(f.shape[0], P, P))
            continue
        bh, bw = rh / P, rw / P
        ys = y1 + (torch.arange(P * gh, device=f.device) // gh).float() * bh + ((torch.arange(P * gh, device=f.device) % gh).float() + 0.5) * (bh / gh)
        xs = x1 + (torch.arange(P * gw, device=f.device) // gw).float() * bw + ((torch.arange(P * gw, device=f.device) % gw).float() + 0.5) * (bw / gw)
        v = _bilinear(f, ys, xs)                                           # [C, P * gh, P * gw]
        outs.append(v.view(-1, P, gh, P, gw).mean((2, 4)))
    return torch.stack(outs).to(feats[0].dtype)


def roi_align_batched(feats, strides, rois, out_size, finest_scale=56, sampling_ratio=0):
    """detector.roi_align's signature (rois [B, n, 4]) on the PyTorch statement: for the CPU tests of the head stand-ins."""
    B, n, _ = rois.shape
    bidx = torch.arange(B, device=rois.device, dtype=rois.dtype)[:, None, None].expand(B, n, 1)
    return roi_align_fpn(list(feats), strides, torch.cat([bidx, rois], -1).reshape(B * n, 5), out_size, sampling_ratio, True, finest_scale)
