"""RoIAlign over the FPN pyramid (csrc/pswin_roi.hip through the C ABI: ops.roi_align_fpn -> pswin_roi_align_fwd / _bwd) against the
plain PyTorch statement of the operator in tests/_roi_ref.py.  The operator is mmcv.ops.RoIAlign with the reference config's
sampling_ratio = 0 (configs/_base_/models/mask_rcnn_swin_fpn.py:46, 65); mmcv.ops is not in the reference tree, so parity with the
REFERENCE is unpinned -- these tests pin the kernels to the published definition."""
import pytest
import torch

import _roi_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
STRIDES = (4, 8, 16, 32)


def _pyramid(B, C, H, W, dtype, seed):
    g = torch.Generator("cpu").manual_seed(seed)
    return [torch.randn(B, C, H // s, W // s, generator=g).to(DEV, dtype) for s in STRIDES]


def _rois(B, H, W, n, seed):
    """Image-pixel RoIs covering the edge cases: every pyramid level, elongated boxes (large adaptive grids), boxes hanging over
    the image border, boxes entirely outside, degenerate (zero-area and inverted) boxes."""
    g = torch.Generator("cpu").manual_seed(seed)
    c = torch.rand(n, 2, generator=g) * torch.tensor([W, H])
    size = torch.exp(torch.rand(n, 2, generator=g) * 5.0 + 1.0)                 # 2.7 .. 400 px per side, aspect ratios up to 1:150
    boxes = torch.cat([c - size / 2, c + size / 2], 1)
    boxes[0] = torch.tensor([-40., -30., 25., 18.])                             # over the top-left corner
    boxes[1] = torch.tensor([W - 10., H - 6., W + 90., H + 70.])                # over the bottom-right corner
    boxes[2] = torch.tensor([W + 50., 10., W + 80., 40.])                       # outside: every sample beyond x > W
    boxes[3] = torch.tensor([30., 40., 30., 40.])                               # zero area
    boxes[4] = torch.tensor([60., 50., 50., 45.])                               # inverted
    boxes[5] = torch.tensor([0., 0., float(W), float(H)])                       # the whole image
    boxes[6] = torch.tensor([-300., -200., 500., 400.])                         # far larger than the image: the coarsest level, mostly outside
    b = torch.randint(0, B, (n, 1), generator=g).float()
    return torch.cat([b, boxes], 1).to(DEV)


@pytest.mark.parametrize("dtype,P,C", [(torch.float32, 7, 256), (torch.bfloat16, 7, 256), (torch.bfloat16, 14, 256), (torch.float32, 3, 64),
                                       (torch.bfloat16, 2, 128), (torch.bfloat16, 5, 512)])
def test_roi_align_forward_matches_the_definition(ops, dtype, P, C):
    B, H, W = 2, 128, 256
    feats = _pyramid(B, C, H, W, dtype, 1)
    rois = _rois(B, H, W, 96, 2)
    got = ops.roi_align_fpn(feats, STRIDES, rois, P)
    want = _roi_ref.roi_align_fpn(feats, STRIDES, rois, P)
    assert got.shape == want.shape == (96, C, P, P)
    lv = _roi_ref.map_roi_levels(rois, 4)
    assert set(lv.tolist()) == {0, 1, 2, 3}
    tol = 2e-5 if dtype == torch.float32 else 1.6e-2                            # bf16: one rounding of the O(1) output
    assert torch.allclose(got.float(), want.float(), rtol=tol, atol=tol), (got.float() - want.float()).abs().max()
    assert torch.equal(got[3], torch.zeros_like(got[3])) and torch.equal(got[2], torch.zeros_like(got[2]))   # degenerate / outside
    # fixed sampling grids use the same kernels
    for sr in (1, 2):
        a = ops.roi_align_fpn(feats, STRIDES, rois, P, sampling_ratio=sr)
        b = _roi_ref.roi_align_fpn(feats, STRIDES, rois, P, sampling_ratio=sr)
        assert torch.allclose(a.float(), b.float(), rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype,P,sr", [(torch.float32, 7, 0), (torch.bfloat16, 14, 0), (torch.float32, 7, 2), (torch.float32, 2, 0)])
def test_roi_align_backward_is_the_adjoint(ops, dtype, P, sr):
    """one atomic per touched cell (separable row / column weights) for footprints up to 16 x 16 cells, the per-sample path beyond
    (P = 2 on 400-pixel boxes at stride 4 .. 8; fixed 2 x 2 grids over large bins: samples further than a cell apart)"""
    B, H, W, C = 2, 128, 256, 256
    feats = [f.requires_grad_(True) for f in _pyramid(B, C, H, W, dtype, 3)]
    rois = _rois(B, H, W, 64, 4)
    w = torch.randn(64, C, P, P, device=DEV)
    (ops.roi_align_fpn(feats, STRIDES, rois, P, sampling_ratio=sr).float() * w).sum().backward()
    got = [f.grad.float().clone() for f in feats]
    ref_feats = [f.detach().float().requires_grad_(True) for f in feats]
    (_roi_ref.roi_align_fpn(ref_feats, STRIDES, rois, P, sampling_ratio=sr) * (w.to(dtype).float() if dtype != torch.float32 else w)).sum().backward()
    used = set(_roi_ref.map_roi_levels(rois, 4).tolist())
    assert len(used) >= 3
    for l, (g, r) in enumerate(zip(got, ref_feats)):
        if l not in used:                                                       # a level no RoI maps to: zero gradient, not None
            assert r.grad is None or float(r.grad.abs().max()) == 0.0
            assert float(g.abs().max()) == 0.0
            continue
        scale = float(r.grad.abs().max())
        tol = 1e-4 if dtype == torch.float32 else 1e-2                          # f32 atomics in arrival order; bf16: dout and dfeat rounded once each
        assert g.shape == r.grad.shape and scale > 0
        assert float((g - r.grad).abs().max()) <= tol * scale, (l, float((g - r.grad).abs().max()), scale)


def test_roi_align_rejects_what_it_is_not_built_for(ops):
    from panoswintransformerobjectdetection_amd._lib import PswinError
    feats = _pyramid(1, 96, 64, 128, torch.bfloat16, 5)
    with pytest.raises(PswinError):
        ops.roi_align_fpn(feats, STRIDES, _rois(1, 64, 128, 8, 6), 7)
    with pytest.raises(PswinError):
        ops.roi_align_fpn([f.cpu() for f in _pyramid(1, 256, 64, 128, torch.float32, 5)], STRIDES, _rois(1, 64, 128, 8, 6).cpu(), 7)


def _greedy(boxes, thr):
    from panoswintransformerobjectdetection_amd.detector import box_iou
    iou = box_iou(boxes, boxes).cpu()
    keep = torch.ones(boxes.shape[0], dtype=torch.bool)
    for i in range(boxes.shape[0]):
        if keep[i]:
            keep &= ~((iou[i] > thr) & (torch.arange(boxes.shape[0]) > i))
    return keep


@pytest.mark.parametrize("thr", [0.7, 0.3])
def test_nms_groups_is_the_sequential_greedy_rule(ops, thr):
    """pswin_nms_groups (one workgroup per list, 64-row bit-mask chunks, one wave walking them) against the sequential rule on the same
    IoU definition: lists of 2000, 1536, 384, 65, 64, 1 and 0 boxes in one launch, dense clusters (long suppression chains), duplicates
    (IoU exactly 1), degenerate boxes."""
    g = torch.Generator("cpu").manual_seed(11)
    lists = []
    for n in (2000, 1536, 384, 65, 64, 1, 0):
        c = torch.rand(n, 2, generator=g) * torch.tensor([1024., 512.]) * 0.25        # crowded: many overlaps
        wh = torch.exp(torch.rand(n, 2, generator=g) * 3.0 + 1.5)
        b = torch.cat([c - wh / 2, c + wh / 2], 1)
        if n >= 64:
            b[5] = b[3]                                                                 # duplicate
            b[7] = torch.tensor([10., 10., 10., 30.])                                   # zero width
            b[9, 2:] = b[9, :2] - 1.0                                                   # inverted
            b[20:60] = b[20:60] * 0 + b[20] + torch.arange(40)[:, None] * 0.5           # a chain of slightly shifted boxes
        lists.append(b.to(DEV))
    got = ops.nms_groups(lists, thr)
    for b, k in zip(lists, got):
        assert k.dtype == torch.bool and k.shape == (b.shape[0],)
        assert torch.equal(k.cpu(), _greedy(b.float(), thr)), (b.shape[0], int(k.sum()))
    with pytest.raises(Exception):
        ops.nms_groups([torch.zeros(2049, 4, device=DEV)], thr)
