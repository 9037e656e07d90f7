"""The minimal Mask R-CNN heads around the backbone (panoswintransformerobjectdetection_amd/detector.py).
Parity with the reference's heads is UNPINNED (mmcv.ops is not in the reference tree, no head fixture exists); these
tests check the pieces against their definitions and that a training step produces finite losses and gradients.  CPU."""
import torch

from panoswintransformerobjectdetection_amd import detector as det

import _roi_ref


def _greedy_nms(boxes, thr):
    keep = []
    for j in range(boxes.shape[0]):
        if all(det.box_iou(boxes[i:i + 1], boxes[j:j + 1]).item() <= thr for i in keep):
            keep.append(j)
    out = torch.zeros(boxes.shape[0], dtype=torch.bool)
    out[keep] = True
    return out


def test_fixed_point_nms_is_greedy_nms():
    torch.manual_seed(0)
    for n in (1, 17, 200):
        c = torch.rand(n, 2) * 60
        wh = torch.rand(n, 2) * 30 + 4
        boxes = torch.cat([c - wh / 2, c + wh / 2], 1)                # heavily overlapping: long suppression chains
        assert torch.equal(det.nms_keep(boxes, 0.5, iters=32), _greedy_nms(boxes, 0.5))


def test_delta_coder_round_trip_and_anchor_layout():
    torch.manual_seed(1)
    a = torch.tensor([[10., 20., 50., 80.], [0., 0., 16., 16.]])
    g = torch.tensor([[12., 25., 60., 70.], [2., 1., 20., 30.]])
    stds = (0.1, 0.1, 0.2, 0.2)
    back = det.decode_deltas(a, det.encode_deltas(a, g, stds), stds, (1000, 1000))
    assert torch.allclose(back, g, atol=1e-3)
    anc = det.make_anchors([(2, 3)], [4], "cpu")[0]
    assert anc.shape == (2 * 3 * 3, 4)
    wh = anc[:, 2:] - anc[:, :2]
    assert torch.allclose(wh[:, 0] * wh[:, 1], torch.full((18,), 32.0 * 32.0), rtol=1e-5)      # scale 8 x stride 4, every ratio
    assert torch.allclose((wh[:3, 1] / wh[:3, 0]), torch.tensor([0.5, 1.0, 2.0]), rtol=1e-5)
    assert torch.allclose((anc[3:6, :2] + anc[3:6, 2:]) / 2, torch.tensor([[4.0, 0.0]] * 3))    # location-major, x fastest


def test_roi_align_statement_reproduces_a_linear_ramp():
    """The PyTorch statement of RoIAlign (tests/_roi_ref.py, the checker of the HIP kernels): bilinear sampling is exact on a linear
    ramp, so every bin of an interior RoI returns the ramp at the bin centre (aligned = True: pixel centres at integer + 0.5)."""
    B, C, H, W = 2, 3, 16, 32
    ys, xs = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    f = torch.stack([xs, ys, torch.ones_like(xs)])[None].repeat(B, 1, 1, 1)                      # channels: x, y, 1 at pixel centres
    feats = [f, f[:, :, ::2, ::2], f[:, :, ::4, ::4], f[:, :, ::8, ::8]]
    rois = torch.tensor([[[8., 8., 40., 24.], [20., 12., 60., 44.]]] * B)                         # image pixels, stride 4 level
    out = _roi_ref.roi_align_batched(feats, (4, 8, 16, 32), rois, 2, finest_scale=56)
    assert out.shape == (B * 2, C, 2, 2)
    x1, y1, x2, y2 = 2.0, 2.0, 10.0, 6.0
    want_x = torch.tensor([x1 + (x2 - x1) * 0.25, x1 + (x2 - x1) * 0.75]) - 0.5
    want_y = torch.tensor([y1 + (y2 - y1) * 0.25, y1 + (y2 - y1) * 0.75]) - 0.5
    assert torch.allclose(out[0, 0, 0], want_x, atol=1e-4) and torch.allclose(out[0, 1, :, 0], want_y, atol=1e-4)
    assert torch.allclose(out[:, 2], torch.ones(B * 2, 2, 2), atol=1e-5)
    # adaptive grid: an 8 x 4-cell RoI pooled to 2 x 2 takes ceil(4 / 2) x ceil(8 / 2) = 2 x 4 samples per bin; a fixed 1 x 1 grid
    # differs on a non-linear map, the adaptive one equals the mean over its sub-bin centres
    g = torch.rand(1, 1, 16, 32)
    r5 = torch.tensor([[0., 8., 8., 40., 24.]])
    a = _roi_ref.roi_align_fpn([g], (4,), r5, 2, sampling_ratio=0)
    b = _roi_ref.roi_align_fpn([g], (4,), r5, 2, sampling_ratio=1)
    assert not torch.allclose(a, b)


def test_heads_training_step_is_finite_and_reaches_every_parameter():
    torch.manual_seed(0)
    m = det.MiniMaskRCNN(dict(embed_dim=96, depths=[2, 2, 2, 2], num_heads=[3, 6, 12, 24], ape=True), num_classes=80)
    m.roi_align = _roi_ref.roi_align_batched          # CPU: the PyTorch statement stands in for the HIP operator (GPU: tests/test_roi_gpu.py)
    B, H, W = 2, 128, 256
    feats = [torch.randn(B, c, H // s, W // s, requires_grad=True) for c, s in zip((96, 192, 384, 768), (4, 8, 16, 32))]
    tg = det.synthetic_targets(B, H, W, "cpu")
    assert all(t["boxes"].shape[0] == t["labels"].shape[0] == t["masks"].shape[0] for t in tg)
    losses = m.heads_loss(feats, tg, (H, W))
    assert set(losses) == {"loss_rpn_cls", "loss_rpn_bbox", "loss_cls", "loss_bbox", "loss_mask"}
    total = sum(losses.values())
    assert torch.isfinite(total)
    total.backward()
    assert all(f.grad is not None and torch.isfinite(f.grad).all() and f.grad.abs().sum() > 0 for f in feats)
    for k, p in m.named_parameters():
        if not k.startswith("backbone."):
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
