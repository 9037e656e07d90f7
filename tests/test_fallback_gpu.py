"""The options outside the kernels' specialisation end to end on the GPU: a window_size = 5 / head_dim = 16 model
-- stem, position encoder, PatchMerging and output norms on the HIP kernels, window blocks as plain torch ops (fallback.py) -- against the
CPU oracle with the same weights, forward and backward; and dropout > 0 runs."""
import warnings

import pytest
import torch

import panoswin_oracle as po
from _util import ZERO_GRAD_KEYS, build_filled, run_and_collect

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CFG5 = dict(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[2, 4, 8, 16], window_size=5, ape=True, drop_path_rate=0.0)


@pytest.mark.parametrize("pano", [True, False])
def test_window_5_head_dim_16_model_matches_the_oracle(pano):
    from panoswintransformerobjectdetection_amd import SimplePanoSwinTransformer
    shape, tag = (2, 3, 64, 128), "fallback5"
    ref = run_and_collect(build_filled(po.SimplePanoSwinTransformerOracle, CFG5, pano, tag), shape, tag)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = build_filled(SimplePanoSwinTransformer, CFG5, pano, tag).to(DEV)
    assert all(b.generic for layer in m.layers for b in layer.blocks)
    got = run_and_collect(m, shape, tag, device=DEV)
    assert ref.keys() == got.keys()
    for k, r in ref.items():
        if any(z in k for z in ZERO_GRAD_KEYS):              # true gradient identically zero: rounding noise on both sides
            continue
        if k == "dx_sub":      # through the two train-mode BatchNorms of the fp32 (MIOpen) stem: isolated near-cancellation elements, so the
            err = float((got[k] - r).norm() / r.norm())               # input gradient is compared in the 2-norm (as the T-size test does)
            assert err < 5e-3, (k, err)
            continue
        scale = float(r.abs().max()) + 1e-12
        err = float((got[k] - r).abs().max()) / scale
        # the blocks themselves are pinned at 1e-5 on the CPU (tests/test_fallback.py); here fp32 GPU GEMMs / MIOpen convolutions against
        # the CPU, and the stem's parameter gradients sit behind two train-mode BatchNorms that amplify summation-order noise
        bound = 2e-3 if k.startswith("out") else (2e-2 if "patch_embed" in k else 5e-3)
        assert err < bound, (k, err)


def test_dropout_model_trains():
    from panoswintransformerobjectdetection_amd import SimplePanoSwinTransformer
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = SimplePanoSwinTransformer(embed_dim=32, depths=[2, 2], num_heads=[1, 2], ape=True, drop_rate=0.1, attn_drop_rate=0.1, out_indices=(0, 1)).to(DEV)
    m.init_weights(None)
    x = torch.randn(2, 3, 64, 128, device=DEV)
    m.train()
    a, b = m(x), m(x)
    assert not torch.equal(a[0], b[0])                       # dropout draws
    sum(o.square().mean() for o in a).backward()
    assert all(p.grad is None or bool(torch.isfinite(p.grad).all()) for p in m.parameters())
    m.eval()
    with torch.no_grad():
        c, d = m(x), m(x)
    assert torch.equal(c[0], d[0]) and torch.equal(c[1], d[1])
