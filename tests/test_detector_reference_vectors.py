"""detector.py's assignment, box decoding and anchor layout against the vectors the REFERENCE's own unit tests hold (VERDICT r3
"Missing 1" / item 7): tests/golden/detector_reference_vectors.json transcribes the inputs and expected outputs of
/root/reference/tests/test_utils/test_assigner.py:14-151, test_coder.py:26-60 and test_anchor.py:22-40 (data, no source text).
CPU only.  Everything else in detector.py stays "parity unpinned" (mmcv.ops is not in the reference tree)."""
import json
import os

import pytest
import torch

from panoswintransformerobjectdetection_amd import detector as D

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "detector_reference_vectors.json")) as f:
    VEC = json.load(f)


def _boxes(rows):
    return torch.tensor(rows, dtype=torch.float32).reshape(-1, 4)


@pytest.mark.parametrize("case", VEC["max_iou_assigner"], ids=lambda c: c["source"].split(":")[-1])
def test_max_iou_assigner_reproduces_the_reference_vectors(case):
    ign = _boxes(case["gt_bboxes_ignore"]) if "gt_bboxes_ignore" in case else None
    got = D.max_iou_assign(_boxes(case["bboxes"]), _boxes(case["gt_bboxes"]), case["pos_iou_thr"], case["neg_iou_thr"],
                           min_pos_iou=0.0, match_low_quality=True, gt_bboxes_ignore=ign,
                           ignore_iof_thr=case.get("ignore_iof_thr", -1.0), ignore_wrt_candidates=case.get("ignore_wrt_candidates", True))
    assert got.dtype == torch.long and got.tolist() == case["expected_gt_inds"]
    if "gt_labels" in case and len(case["expected_gt_inds"]):      # the labels the reference derives from gt_inds (:199-207)
        labels = torch.tensor(case["gt_labels"])
        pos = got > 0
        assert labels[got[pos] - 1].tolist() == [case["gt_labels"][i - 1] for i in case["expected_gt_inds"] if i > 0]


def test_low_quality_matching_lets_a_later_gt_overwrite_an_earlier_one_like_the_sequential_loop():
    """max_iou_assigner.py:182-197 walks the gts in order; the vectorised form must give the LAST matching gt (and respect min_pos_iou)."""
    torch.manual_seed(0)
    for trial in range(20):
        n, g = 40, 6
        xy = torch.rand(n + g, 2) * 40
        wh = torch.rand(n + g, 2) * 30 + 2
        allb = torch.cat([xy, xy + wh], 1)
        b, gt = allb[:n], allb[n:]
        if trial % 3 == 0:
            gt[1] = gt[0]                                             # exact ties between two gts
        got = D.max_iou_assign(b, gt, 0.7, 0.3, 0.3, True)
        ov = D.box_iou(gt, b)
        best, arg = ov.max(0)
        want = torch.full((n,), -1, dtype=torch.long)
        want[(best >= 0) & (best < 0.3)] = 0
        want[best >= 0.7] = arg[best >= 0.7] + 1
        gbest = ov.max(1)[0]
        for i in range(g):                                           # the reference's loop, verbatim semantics
            if gbest[i] >= 0.3:
                want[ov[i] == gbest[i]] = i + 1
        assert torch.equal(got, want), trial
        assert torch.equal(D.max_iou_assign(b, gt, 0.5, 0.5, 0.5, False), torch.where(best >= 0.5, arg + 1, torch.zeros_like(arg)))


def test_delta_xywh_decode_reproduces_the_reference_vectors():
    c = VEC["delta_xywh_bbox_coder_decode"]
    out = D.decode_deltas(_boxes(c["rois"]), _boxes(c["deltas"]), tuple(c["stds"]), tuple(c["max_shape"]))
    assert torch.allclose(out, _boxes(c["expected"]), atol=c["atol"])
    assert D.decode_deltas(torch.zeros(0, 4), torch.zeros(0, 4), (1.0, 1.0, 1.0, 1.0), (32, 32)).shape == (0, 4)     # test_coder.py:57-60
    # encode is the inverse on boxes inside the clip range (delta_xywh_bbox_coder.py:100-129)
    src = torch.tensor([[2.0, 3.0, 12.0, 9.0], [5.0, 5.0, 25.0, 30.0]])
    dst = torch.tensor([[1.0, 4.0, 14.0, 12.0], [6.0, 2.0, 20.0, 28.0]])
    for stds in ((1.0, 1.0, 1.0, 1.0), (0.1, 0.1, 0.2, 0.2)):
        assert torch.allclose(D.decode_deltas(src, D.encode_deltas(src, dst, stds), stds, (64, 64)), dst, atol=1e-4)


@pytest.mark.parametrize("case", VEC["anchor_generator_grid"], ids=lambda c: c["source"].split(":")[-1])
def test_anchor_grid_layout_reproduces_the_reference_vectors(case):
    D._ANCHORS.clear()
    got = D.make_anchors([tuple(s) for s in case["featmap_sizes"]], case["strides"], "cpu", scale=case["scales"][0], ratios=tuple(case["ratios"]))
    assert len(got) == 1 and torch.equal(got[0], _boxes(case["expected"]))


def test_anchor_base_shapes_of_the_mask_rcnn_config():
    """AnchorGenerator(scales=[8], ratios=[0.5, 1, 2], strides=[4 .. 64]) (configs/_base_/models/mask_rcnn_swin_fpn.py:24-28): per cell three
    anchors of area (8 s)^2 with h / w = ratio, ratio-major, centred on the cell's corner (center_offset 0)."""
    D._ANCHORS.clear()
    a = D.make_anchors([(2, 3)], [16], "cpu")[0].view(2, 3, 3, 4)
    w, h = a[..., 2] - a[..., 0], a[..., 3] - a[..., 1]
    assert torch.allclose(w * h, torch.full_like(w, 128.0 ** 2), rtol=1e-5)
    assert torch.allclose((h / w)[0, 0], torch.tensor([0.5, 1.0, 2.0]), rtol=1e-5)
    cx, cy = (a[..., 0] + a[..., 2]) / 2, (a[..., 1] + a[..., 3]) / 2
    assert torch.equal(cx[:, :, 0], torch.tensor([[0.0, 16.0, 32.0]] * 2)) and torch.equal(cy[:, :, 0], torch.tensor([[0.0] * 3, [16.0] * 3]))
