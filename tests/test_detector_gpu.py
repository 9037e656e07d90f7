"""BASELINE.json configs[2] on the MI355X box: PanoSwin backbone (HIP) + the minimal Mask R-CNN head stack, one training
step.  Head parity is unpinned (detector.py); what is asserted: finite losses, gradients for every parameter, and that the
backbone's parameter gradients through the detector equal the backbone-only backward pass fed with the same feature-map
gradients (the two-graph arrangement bench.py --config maskrcnn replays)."""
import pytest
import torch

from _util import TINY

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("cd", [torch.float32, torch.bfloat16])
def test_mask_rcnn_step_and_backbone_gradient_hand_off(cd):
    from panoswintransformerobjectdetection_amd.detector import MiniMaskRCNN, synthetic_targets
    torch.manual_seed(0)
    m = MiniMaskRCNN(dict(TINY, compute_dtype=cd), num_classes=80).to(DEV).train()
    m.backbone.init_weights(None)
    B, H, W = 2, 128, 256
    x = torch.randn(B, 3, H, W, device=DEV)
    tg = synthetic_targets(B, H, W, DEV)

    torch.manual_seed(7)                                       # the samplers draw random keys: same draws in both runs
    losses = m.forward_train(x, tg)
    total = sum(losses.values())
    assert torch.isfinite(total) and all(torch.isfinite(v) and v >= 0 for v in losses.values())
    total.backward()
    ref = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    assert all(torch.isfinite(g).all() for g in ref.values())
    assert sum(float(g.abs().sum()) > 0 for k, g in ref.items() if k.startswith("backbone.")) > 100

    for p in m.parameters():
        p.grad = None
    outs = m.backbone(x)
    feats = [o.detach().requires_grad_(True) for o in outs]
    torch.manual_seed(7)
    sum(m.heads_loss(feats, tg, (H, W)).values()).backward()
    torch.autograd.backward(outs, [f.grad for f in feats])
    for k, p in m.named_parameters():
        tol = 1e-5 if cd == torch.float32 else 1e-2
        assert torch.allclose(p.grad, ref[k], rtol=tol, atol=tol * float(ref[k].abs().max()) + 1e-12), k
