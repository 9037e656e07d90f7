"""A minimal Mask R-CNN training step around the MI355X PanoSwin backbone (BASELINE.json configs[2]).

Scope: the CALLER of the hot path, built only far enough to time an end-to-end detector step with the backbone's share in
it.  mmcv / mmdet / torchvision are not installed, so this is a self-contained pure-PyTorch restatement of ONE pipeline with
the reference's configuration numbers (configs/_base_/models/mask_rcnn_swin_fpn.py:21-115):

  FPN (mmdet/models/necks/fpn.py: laterals 1x1, top-down nearest upsampling, 3x3 output convs, 5th level by stride-2
  subsampling) -> RPNHead (3 anchors: scale 8, ratios 0.5 / 1 / 2, strides 4..64; MaxIoU assigner 0.7 / 0.3 / 0.3, random
  sampler 256 @ 0.5, sigmoid cross-entropy + L1 on deltas) -> proposals (nms_pre 2000 per level, NMS 0.7, 1000 per image)
  -> StandardRoIHead (MaxIoU 0.5, ground truth added as proposals, random sampler 512 @ 0.25; RoIAlign 7 -> Shared2FC 1024
  -> 80 classes, cross-entropy + L1 with stds 0.1 / 0.2; RoIAlign 14 -> 4 convs + deconv -> 28 x 28 masks, BCE on the class
  channel), called as TwoStageDetector.forward_train does (mmdet/models/detectors/two_stage.py:116-175).

PARITY: unpinned.  The reference tree does not contain mmcv.ops (RoIAlign, NMS CUDA sources) nor a fixture for any head,
so nothing here is checked against reference outputs.  RoIAlign is the published operator with the config's sampling_ratio = 0
(adaptive grid) as hand-written HIP kernels, forward and backward (csrc/pswin_roi.hip, checked against a plain PyTorch statement
of the definition in tests/); NMS is the greedy rule: on the GPU one HIP launch per image (nms_keep_groups -> pswin_nms_groups), on the CPU a fixed-point iteration (nms_keep).  The other
head operators are ordinary PyTorch-ROCm operators (MIOpen / hipBLASLt, bf16 autocast).
"""
import math

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .backbone import SimplePanoSwinTransformer


# ------------------------------------------------------------------------------------------------------------------------
# boxes
# ------------------------------------------------------------------------------------------------------------------------
def box_iou(a, b):
    """[N, 4] x [M, 4] (x1, y1, x2, y2) -> [N, M]"""
    area_a = (a[:, 2] - a[:, 0]).clamp(min=0) * (a[:, 3] - a[:, 1]).clamp(min=0)
    area_b = (b[:, 2] - b[:, 0]).clamp(min=0) * (b[:, 3] - b[:, 1]).clamp(min=0)
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[:, None] + area_b[None] - inter).clamp(min=1e-6)


def box_iof(a, b):
    """intersection over the area of the FIRST box (mmdet bbox_overlaps(mode='iof')): [N, 4] x [M, 4] -> [N, M]"""
    area_a = (a[:, 2] - a[:, 0]).clamp(min=0) * (a[:, 3] - a[:, 1]).clamp(min=0)
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    return wh[..., 0] * wh[..., 1] / area_a[:, None].clamp(min=1e-6)


def max_iou_assign(bboxes, gt_bboxes, pos_iou_thr, neg_iou_thr, min_pos_iou=0.0, match_low_quality=True, gt_bboxes_ignore=None,
                   ignore_iof_thr=-1.0, ignore_wrt_candidates=True):
    """MaxIoUAssigner.assign (mmdet/core/bbox/assigners/max_iou_assigner.py:85-212; gt_max_assign_all=True): long [N] with
    -1 = ignore, 0 = negative, i + 1 = positive matched to gt i.  Pinned by the reference's own vectors
    (tests/test_utils/test_assigner.py:14-151 -> tests/golden/detector_reference_vectors.json).  No host synchronisation and static
    shapes for non-empty inputs (the step is captured in a hipGraph): the reference's sequential low-quality loop, in which a later
    gt overwrites an earlier one, is the maximum over the matching gt indices."""
    N, G = bboxes.shape[0], gt_bboxes.shape[0]
    inds = torch.full((N,), -1, dtype=torch.long, device=bboxes.device)
    if N == 0 or G == 0:
        return inds.zero_() if G == 0 else inds                                                   # :147-153
    overlaps = box_iou(gt_bboxes, bboxes)                                                         # [G, N]  (:105)
    if ignore_iof_thr > 0 and gt_bboxes_ignore is not None and gt_bboxes_ignore.numel() > 0:    # :107-117
        if ignore_wrt_candidates:
            ign = box_iof(bboxes, gt_bboxes_ignore).max(1)[0]
        else:
            ign = box_iof(gt_bboxes_ignore, bboxes).max(0)[0]
        overlaps = torch.where((ign > ignore_iof_thr)[None], overlaps.new_full((), -1.0), overlaps)
    best, arg = overlaps.max(0)
    inds = torch.where((best >= 0) & (best < neg_iou_thr), torch.zeros_like(inds), inds)          # :170-172
    inds = torch.where(best >= pos_iou_thr, arg + 1, inds)                                        # :179-180
    if match_low_quality:                                                                         # :182-197
        gbest = overlaps.max(1)[0]
        hit = (overlaps == gbest[:, None]) & (gbest[:, None] >= min_pos_iou)
        last = (hit.long() * torch.arange(1, G + 1, device=bboxes.device)[:, None]).max(0)[0]
        inds = torch.where(last > 0, last, inds)
    return inds


_CONSTS = {}


def _const(values, like):
    """A small constant tensor on `like`'s device, built once (a host-to-device copy per call would also break hipGraph capture)."""
    key = (tuple(float(v) for v in values), str(like.device), like.dtype)
    if key not in _CONSTS:
        _CONSTS[key] = torch.tensor(key[0], device=like.device, dtype=like.dtype)
    return _CONSTS[key]


def encode_deltas(src, dst, stds):
    """DeltaXYWHBBoxCoder.encode (means 0)"""
    sw, sh = (src[:, 2] - src[:, 0]).clamp(min=1e-3), (src[:, 3] - src[:, 1]).clamp(min=1e-3)
    dw, dh = (dst[:, 2] - dst[:, 0]).clamp(min=1e-3), (dst[:, 3] - dst[:, 1]).clamp(min=1e-3)
    sx, sy = (src[:, 0] + src[:, 2]) * 0.5, (src[:, 1] + src[:, 3]) * 0.5
    dx, dy = (dst[:, 0] + dst[:, 2]) * 0.5, (dst[:, 1] + dst[:, 3]) * 0.5
    d = torch.stack([(dx - sx) / sw, (dy - sy) / sh, torch.log(dw / sw), torch.log(dh / sh)], 1)
    return d / _const(stds, d)


def decode_deltas(src, deltas, stds, max_shape):
    d = deltas * _const(stds, deltas)
    sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
    sx, sy = (src[:, 0] + src[:, 2]) * 0.5, (src[:, 1] + src[:, 3]) * 0.5
    clip = abs(math.log(16 / 1000))
    w, h = sw * d[:, 2].clamp(-clip, clip).exp(), sh * d[:, 3].clamp(-clip, clip).exp()
    x, y = sx + sw * d[:, 0], sy + sh * d[:, 1]
    H, W = max_shape
    return torch.stack([(x - w * 0.5).clamp(0, W), (y - h * 0.5).clamp(0, H), (x + w * 0.5).clamp(0, W), (y + h * 0.5).clamp(0, H)], 1)


def _topk_stable(scores, k):
    """The k largest scores and their indices, ties broken by index (stable sort).  torch.topk picks among EQUAL scores in an order
    that depends on the timing of its atomics: with bf16 RPN logits (thousands of exactly equal values at initialisation) two passes
    over bit-identical feature maps then keep different proposals -- same losses to six digits, feature-map gradients 11 % apart
    (seen between replays of one captured step).  mmdet's own top-k has the same property; a stand-in that is replayed from a hipGraph
    and compared with an eager step should not."""
    s, i = torch.sort(scores, descending=True, stable=True)
    return s[:k], i[:k]


def nms_keep(boxes, iou_thr, iters=12):
    """Greedy NMS on score-sorted boxes as a fixed point: keep[j] = not any(i < j: keep[i] and IoU(i, j) > thr).  Starting
    from "keep all", every sweep fixes at least one more level of the suppression chains; `iters` sweeps of one [n, n]
    mask-vector product each, no host synchronisation (the step is captured in a hipGraph: a static sweep count).  Exact
    when no suppression chain is deeper than `iters`; APPROXIMATE beyond that (the default 12 is not checked for convergence
    in the step; tests/test_detector_heads.py compares 32 sweeps with the sequential rule)."""
    over = torch.triu(box_iou(boxes, boxes) > iou_thr, diagonal=1).float()
    keep = torch.ones(boxes.shape[0], device=boxes.device)
    for _ in range(iters):
        keep = (keep @ over == 0).float()
    return keep.bool()


def nms_keep_groups(box_list, iou_thr):
    """nms_keep for several score-sorted box lists.  On the GPU: the exact greedy rule, all lists in one HIP launch (ops.nms_groups ->
    pswin_nms_groups; <= 2048 boxes per list); on the CPU (the definition tests): the fixed-point form above, list by list."""
    if box_list and box_list[0].is_cuda and max(b.shape[0] for b in box_list) <= 2048:
        from . import ops
        return ops.nms_groups(box_list, iou_thr)
    return [nms_keep(b, iou_thr) for b in box_list]


# ------------------------------------------------------------------------------------------------------------------------
# neck and heads
# ------------------------------------------------------------------------------------------------------------------------
class _AddConvBias(torch.autograd.Function):
    """y = x + bias[None, :, None, None] whose bias gradient is a fixed-order column sum through the C ABI (ops.colsum on the
    channels-last rows).  A convolution's own bias gradient is one of the framework's two-pass global reductions
    (MIOpen ConvolutionBackwardBias / at::sum): 29 launches of ~21 us per step here, and the kind of reduction that returns stale
    results when a captured hipGraph is replayed (pitfalls) -- the head stand-ins of the step ARE replayed."""

    @staticmethod
    def forward(ctx, x, bias):
        ctx.bias_dtype = bias.dtype
        return x + bias.to(x.dtype).view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, dy):
        if dy.is_cuda:
            from . import ops
            db = ops.colsum_channels(dy)                # every channel count through the C ABI (narrow rows are zero-padded)
        else:
            db = dy.float().sum((0, 2, 3))              # CPU: the definition tests only
        return dy, db.to(ctx.bias_dtype)


def conv_bias(m, x):
    """m(x) for an nn.Conv2d / nn.ConvTranspose2d with the bias added (and its gradient summed) by _AddConvBias"""
    if isinstance(m, nn.ConvTranspose2d):
        y = F.conv_transpose2d(x, m.weight, None, m.stride, m.padding, m.output_padding, m.groups, m.dilation)
    else:
        y = F.conv2d(x, m.weight, None, m.stride, m.padding, m.dilation, m.groups)
    return y if m.bias is None else _AddConvBias.apply(y, m.bias)


class FPN(nn.Module):
    def __init__(self, in_channels=(96, 192, 384, 768), out_channels=256, num_outs=5):
        super().__init__()
        self.lateral = nn.ModuleList(nn.Conv2d(c, out_channels, 1) for c in in_channels)
        self.output = nn.ModuleList(nn.Conv2d(out_channels, out_channels, 3, padding=1) for _ in in_channels)
        self.num_outs = num_outs

    def forward(self, feats):
        lat = [conv_bias(l, f) for l, f in zip(self.lateral, feats)]
        for i in range(len(lat) - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode="nearest")
        outs = [conv_bias(o, x) for o, x in zip(self.output, lat)]
        while len(outs) < self.num_outs:
            outs.append(F.max_pool2d(outs[-1], 1, stride=2))
        return outs


class RPNHead(nn.Module):
    def __init__(self, channels=256, num_anchors=3):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)
        self.cls = nn.Conv2d(channels, num_anchors, 1)
        self.reg = nn.Conv2d(channels, num_anchors * 4, 1)

    def forward(self, feats):
        outs = []
        for f in feats:
            t = F.relu(conv_bias(self.conv, f))
            outs.append((conv_bias(self.cls, t), conv_bias(self.reg, t)))
        return outs


_ANCHORS = {}


def make_anchors(shapes, strides, device, scale=8.0, ratios=(0.5, 1.0, 2.0)):
    """AnchorGenerator(scales=[8], ratios=[0.5, 1, 2]).grid_anchors (mmdet/core/anchor/anchor_generator.py; center_offset 0,
    scale_major): per level [H * W * 3, 4], location-major with x fastest (the conv output order).  A stride may be an (x, y) pair; the
    base size is then min(stride), as in the reference.  Layout pinned by tests/test_utils/test_anchor.py:22-40 of the reference
    (tests/golden/detector_reference_vectors.json).  Cached per geometry: constants of the step (and building them copies host data,
    which a hipGraph capture refuses)."""
    strides = [tuple(s) if isinstance(s, (tuple, list)) else (s, s) for s in strides]
    key = (tuple(tuple(int(v) for v in sh) for sh in shapes), tuple(strides), str(device), scale, tuple(ratios))
    if key in _ANCHORS:
        return _ANCHORS[key]
    out = []
    for (H, W), (stx, sty) in zip(shapes, strides):
        s = min(stx, sty)
        r = torch.tensor(ratios, device=device)
        hr, wr = torch.sqrt(r), 1.0 / torch.sqrt(r)
        ws, hs = s * scale * wr, s * scale * hr
        base = torch.stack([-0.5 * ws, -0.5 * hs, 0.5 * ws, 0.5 * hs], 1)                         # centred on the cell corner
        sy, sx = torch.meshgrid(torch.arange(H, device=device) * sty, torch.arange(W, device=device) * stx, indexing="ij")
        shift = torch.stack([sx, sy, sx, sy], -1).reshape(-1, 1, 4).float()
        out.append((shift + base[None]).reshape(-1, 4))
    _ANCHORS[key] = out
    return out


class BBoxHead(nn.Module):
    def __init__(self, channels=256, roi=7, fc=1024, num_classes=80):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(channels * roi * roi, fc), nn.Linear(fc, fc)
        self.cls, self.reg = nn.Linear(fc, num_classes + 1), nn.Linear(fc, num_classes * 4)

    def forward(self, x):
        x = F.relu(self.fc2(F.relu(self.fc1(x.flatten(1)))))
        return self.cls(x), self.reg(x)


class MaskHead(nn.Module):
    def __init__(self, channels=256, num_convs=4, num_classes=80):
        super().__init__()
        self.convs = nn.ModuleList(nn.Conv2d(channels, channels, 3, padding=1) for _ in range(num_convs))
        self.up = nn.ConvTranspose2d(channels, channels, 2, stride=2)
        self.logits = nn.Conv2d(channels, num_classes, 1)

    def forward(self, x):
        for c in self.convs:
            x = F.relu(conv_bias(c, x))
        return conv_bias(self.logits, F.relu(conv_bias(self.up, x)))


def roi_align(feats, strides, rois, out_size, finest_scale=56, sampling_ratio=0):
    """SingleRoIExtractor + RoIAlign(output_size, sampling_ratio = 0: adaptive ceil(roi / output) samples per bin) over 4 FPN levels
    (configs/_base_/models/mask_rcnn_swin_fpn.py:44-48, 63-67).  rois [B, n, 4] in image pixels (the same number per image) ->
    [B * n, C, out, out].  One HIP kernel per direction (ops.roi_align_fpn -> pswin_roi_align_fwd / _bwd): every RoI is sampled on
    the level its scale maps to (finest_scale = 56, as mmdet) and only there."""
    from . import ops
    B, n, _ = rois.shape
    bidx = torch.arange(B, device=rois.device, dtype=rois.dtype)[:, None, None].expand(B, n, 1)
    rois5 = torch.cat([bidx, rois], -1).reshape(B * n, 5)
    return ops.roi_align_fpn(list(feats), strides, rois5, out_size, sampling_ratio, True, finest_scale)


# ------------------------------------------------------------------------------------------------------------------------
# the detector
# ------------------------------------------------------------------------------------------------------------------------
class MiniMaskRCNN(nn.Module):
    """backbone -> FPN -> RPN -> RoI heads with the train_cfg numbers of configs/_base_/models/mask_rcnn_swin_fpn.py.
    `heads_loss(feats, targets)` is everything behind the backbone; `forward_train` = backbone + heads_loss."""

    STRIDES = (4, 8, 16, 32, 64)
    rand_like = staticmethod(torch.rand_like)    # the samplers' random keys (tests substitute a fixed sequence to compare eager and replayed steps)
    roi_align = staticmethod(roi_align)          # the HIP operator; tests of the head stand-ins on the CPU substitute the PyTorch statement

    def __init__(self, backbone_cfg, num_classes=80):
        super().__init__()
        self.backbone = SimplePanoSwinTransformer(**backbone_cfg)
        c = self.backbone.num_features
        self.neck = FPN(c, 256, 5)
        self.rpn = RPNHead(256, 3)
        self.bbox_head = BBoxHead(256, 7, 1024, num_classes)
        self.mask_head = MaskHead(256, 4, num_classes)
        self.num_classes = num_classes
        self.rpn_cfg = dict(pos=0.7, neg=0.3, min_pos=0.3, num=256, pos_fraction=0.5, nms_pre=2000, max_per_img=1000, nms=0.7)
        self.rcnn_cfg = dict(pos=0.5, num=512, pos_fraction=0.25, mask_size=28)
        for part in (self.neck, self.rpn, self.mask_head):
            for m in part.modules():
                if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                    nn.init.normal_(m.weight, std=0.01)
                    nn.init.zeros_(m.bias)
        # The convolutional heads in channels-last memory format (PSWIN_HEADS_CHANNELS_LAST=0: NCHW) (MIOpen's bf16 kernels are NHWC: in NCHW
        # every convolution is wrapped in layout transposes, 150 launches / 0.9 ms per step); needs its own find-db records
        # (tools/miopen_find_heads.sh)
        self.channels_last = os.environ.get("PSWIN_HEADS_CHANNELS_LAST", "1") != "0"
        if self.channels_last:
            for part in (self.neck, self.rpn, self.mask_head):
                part.to(memory_format=torch.channels_last)

    def head_parameters(self):
        bb = {id(p) for p in self.backbone.parameters()}
        return [p for p in self.parameters() if id(p) not in bb]

    # -- RPN ----------------------------------------------------------------------------------------------------------
    def _rpn_losses_and_proposals(self, rpn_outs, anchors, targets, img_hw):
        cfg = self.rpn_cfg
        B = rpn_outs[0][0].shape[0]
        flat_a = torch.cat(anchors, 0)
        cls_all = torch.cat([c.permute(0, 2, 3, 1).reshape(B, -1) for c, _ in rpn_outs], 1).float()          # [B, A]
        reg_all = torch.cat([r.permute(0, 2, 3, 1).reshape(B, -1, 4) for _, r in rpn_outs], 1).float()       # [B, A, 4]
        loss_cls = loss_reg = cls_all.new_zeros(())
        n_pos_max, n_tot = int(cfg["num"] * cfg["pos_fraction"]), cfg["num"]
        proposals = []
        for b in range(B):
            gt = targets[b]["boxes"]
            # MaxIoUAssigner(pos 0.7, neg 0.3, min_pos 0.3, match_low_quality) -- configs/_base_/models/mask_rcnn_swin_fpn.py:79-85
            gt_inds = max_iou_assign(flat_a, gt, cfg["pos"], cfg["neg"], cfg["min_pos"], True)
            label = gt_inds.clamp(max=1).to(flat_a.dtype)                                         # 1 positive, 0 negative, -1 neither
            arg = (gt_inds - 1).clamp(min=0)
            best = label
            # random sampling with static shapes: rank by a random key, positives first
            key = self.rand_like(best)
            pos_rank = torch.argsort(torch.where(label == 1, key, key + 2))[:n_pos_max]
            pos_valid = label[pos_rank] == 1
            neg_rank = torch.argsort(torch.where(label == 0, key, key + 2))[:n_tot]
            n_pos = pos_valid.sum()
            neg_valid = (label[neg_rank] == 0) & (torch.arange(n_tot, device=key.device) < (n_tot - n_pos))
            idx = torch.cat([pos_rank, neg_rank])
            valid = torch.cat([pos_valid, neg_valid]).float()
            tgt = torch.cat([torch.ones_like(pos_valid, dtype=torch.float32), torch.zeros(n_tot, device=key.device)])
            avg = valid.sum().clamp(min=1)
            loss_cls = loss_cls + (F.binary_cross_entropy_with_logits(cls_all[b, idx], tgt, reduction="none") * valid).sum() / avg
            d_t = encode_deltas(flat_a[pos_rank], gt[arg[pos_rank]], (1.0, 1.0, 1.0, 1.0))
            loss_reg = loss_reg + ((reg_all[b, pos_rank] - d_t).abs().sum(1) * pos_valid.float()).sum() / avg
            # proposals (no gradient): per level top nms_pre, decode, NMS, then the best max_per_img over the levels
            with torch.no_grad():
                boxes_l, tops_l, at = [], [], 0
                for a_l in anchors:
                    n = a_l.shape[0]
                    sc = cls_all[b, at:at + n]
                    k = min(cfg["nms_pre"], n)
                    top, ti = _topk_stable(sc, k)
                    boxes_l.append(decode_deltas(a_l[ti], reg_all[b, at:at + n][ti], (1.0, 1.0, 1.0, 1.0), img_hw))
                    tops_l.append(top)
                    at += n
                keeps = nms_keep_groups(boxes_l, cfg["nms"])                   # the levels of one image: one launch on the GPU
                scores_l = [torch.where(kp, top, top.new_full((), -1e4)) for kp, top in zip(keeps, tops_l)]
                bx, sc = torch.cat(boxes_l), torch.cat(scores_l)
                ti = _topk_stable(sc, min(cfg["max_per_img"], sc.numel()))[1]
                proposals.append(bx[ti])
        return loss_cls / B, loss_reg / B, proposals

    # -- RoI heads ------------------------------------------------------------------------------------------------------
    def _roi_losses(self, feats, proposals, targets, img_hw):
        cfg = self.rcnn_cfg
        n_tot, n_pos_max = cfg["num"], int(cfg["num"] * cfg["pos_fraction"])
        rois, labels, reg_t, pos_valid_all, gt_idx_all = [], [], [], [], []
        with torch.no_grad():
            for b, props in enumerate(proposals):
                gt, gl = targets[b]["boxes"], targets[b]["labels"]
                cand = torch.cat([gt, props], 0)                                                  # add_gt_as_proposals
                # MaxIoUAssigner(pos 0.5, neg 0.5, min_pos 0.5, match_low_quality=True) -- mask_rcnn_swin_fpn.py:101-107
                gt_inds = max_iou_assign(cand, gt, cfg["pos"], cfg["pos"], cfg["pos"], True)
                is_pos, arg = gt_inds > 0, (gt_inds - 1).clamp(min=0)
                best = is_pos.float()
                key = self.rand_like(best)
                pos_rank = torch.argsort(torch.where(is_pos, key, key + 2))[:n_pos_max]
                pos_valid = is_pos[pos_rank]
                # the positive slots that found no positive were filled with the lowest-key non-positives (they count as background
                # below): the negatives proper are the NEXT ones in that order, so that no RoI is sampled twice
                filler = n_pos_max - pos_valid.sum()
                neg_order = torch.argsort(torch.where(~is_pos, key, key + 2))
                take = (torch.arange(n_tot - n_pos_max, device=key.device) + filler).clamp(max=neg_order.numel() - 1)
                neg_rank = neg_order[take]
                idx = torch.cat([pos_rank, neg_rank])
                rois.append(cand[idx])
                lab = torch.where(torch.cat([pos_valid, torch.zeros_like(neg_rank, dtype=torch.bool)]), gl[arg[idx]],
                                  torch.full_like(idx, self.num_classes))                          # background = num_classes
                labels.append(lab)
                reg_t.append(encode_deltas(cand[pos_rank], gt[arg[pos_rank]], (0.1, 0.1, 0.2, 0.2)))
                pos_valid_all.append(pos_valid)
                gt_idx_all.append(arg[pos_rank])
        rois_b, labels_c = torch.stack(rois), torch.cat(labels)                                  # [B, n_tot, 4], [B * n_tot]
        B = len(proposals)
        x = self.roi_align(feats[:4], self.STRIDES[:4], rois_b, 7)
        cls, reg = self.bbox_head(x.to(feats[0].dtype))
        loss_cls = F.cross_entropy(cls.float(), labels_c)
        pos_sel = torch.cat([torch.arange(n_pos_max, device=rois_b.device) + b * n_tot for b in range(B)])
        pv = torch.cat(pos_valid_all).float()
        pl = labels_c[pos_sel].clamp(max=self.num_classes - 1)
        ar = torch.arange(pos_sel.numel(), device=reg.device)
        reg_p = reg.float()[pos_sel].view(-1, self.num_classes, 4)[ar, pl]
        loss_bbox = ((reg_p - torch.cat(reg_t)).abs().sum(1) * pv).sum() / (B * n_tot)
        # masks on the positive RoIs (the first n_pos_max of every image)
        xm = self.roi_align(feats[:4], self.STRIDES[:4], rois_b[:, :n_pos_max], 14)
        logits = self.mask_head(xm.to(feats[0].dtype)).float()                                    # [B * P, classes, 28, 28]
        logit_c = logits[ar, pl]
        with torch.no_grad():
            mt = []
            ms = cfg["mask_size"]
            t = (torch.arange(ms, device=rois_b.device, dtype=torch.float32) + 0.5) / ms
            H, W = img_hw
            for b in range(B):
                r = rois_b[b, :n_pos_max]
                gx = (r[:, 0:1] + (r[:, 2:3] - r[:, 0:1]) * t[None]) / W * 2 - 1
                gy = (r[:, 1:2] + (r[:, 3:4] - r[:, 1:2]) * t[None]) / H * 2 - 1
                grid = torch.stack([gx[:, None, :].expand(-1, ms, ms), gy[:, :, None].expand(-1, ms, ms)], -1).reshape(1, -1, ms, 2)
                gm = targets[b]["masks"].float()[None]                                            # [1, G, H, W]: all gt bitmaps as channels
                smp = F.grid_sample(gm, grid, mode="bilinear", padding_mode="zeros", align_corners=False)   # [1, G, P * ms, ms]
                smp = smp[0].view(gm.shape[1], n_pos_max, ms, ms)
                mt.append((smp[gt_idx_all[b], torch.arange(n_pos_max, device=smp.device)] >= 0.5).float())
            mt = torch.cat(mt)
        lm = F.binary_cross_entropy_with_logits(logit_c, mt, reduction="none").mean((1, 2))
        loss_mask = (lm * pv).sum() / pv.sum().clamp(min=1)
        return loss_cls, loss_bbox, loss_mask

    def heads_loss(self, feats, targets, img_hw):
        """Everything behind the backbone: dict of the 5 Mask R-CNN losses (two_stage.py:116-175)."""
        if self.channels_last:
            feats = [f.contiguous(memory_format=torch.channels_last) for f in feats]
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=feats[0].is_cuda):
            fpn = self.neck([f for f in feats])
            rpn_outs = self.rpn(fpn)
        anchors = make_anchors([f.shape[2:] for f in fpn], self.STRIDES, feats[0].device)
        l_rpn_cls, l_rpn_reg, proposals = self._rpn_losses_and_proposals(rpn_outs, anchors, targets, img_hw)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=feats[0].is_cuda):
            l_cls, l_bbox, l_mask = self._roi_losses(fpn, proposals, targets, img_hw)
        return {"loss_rpn_cls": l_rpn_cls, "loss_rpn_bbox": l_rpn_reg, "loss_cls": l_cls, "loss_bbox": l_bbox, "loss_mask": l_mask}

    def forward_train(self, img, targets):
        return self.heads_loss(self.backbone(img), targets, img.shape[2:])


def synthetic_targets(batch, H, W, device, num_classes=80, seed=0):
    """COCO-shaped synthetic targets as the reference's own detector tests build them (tests/test_models/test_forward.py:
    326-392, _demo_mm_inputs: RandomState(0), 1-9 boxes per image from uniform centre / size, labels in [1, classes),
    random bitmap masks); masks here are the box interiors with a random 8 x 8 pattern so that they are learnable shapes
    of the right size rather than 50 % noise at full resolution."""
    import numpy as np
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(batch):
        n = rng.randint(1, 10)
        cx, cy, bw, bh = rng.rand(n, 4).T
        x1, y1 = ((cx * W) - (W * bw / 2)).clip(0, W), ((cy * H) - (H * bh / 2)).clip(0, H)
        x2, y2 = ((cx * W) + (W * bw / 2)).clip(0, W), ((cy * H) + (H * bh / 2)).clip(0, H)
        boxes = np.stack([x1, y1, np.maximum(x2, x1 + 2), np.maximum(y2, y1 + 2)], 1).astype(np.float32)
        labels = rng.randint(1, num_classes, size=n)
        masks = np.zeros((n, H, W), dtype=np.uint8)
        for i, (a, b, c, d) in enumerate(boxes.astype(int)):
            pat = rng.randint(0, 2, (8, 8)).astype(np.uint8)
            hh, ww = max(d - b, 1), max(c - a, 1)
            masks[i, b:b + hh, a:a + ww] = np.kron(pat, np.ones((hh // 8 + 1, ww // 8 + 1), dtype=np.uint8))[:hh, :ww]
        out.append({"boxes": torch.from_numpy(boxes).to(device), "labels": torch.from_numpy(labels).long().to(device),
                    "masks": torch.from_numpy(masks).to(device)})
    return out
