"""The CPU oracle against the LIVE reference (only where /root/reference exists; skipped on the GPU box)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import panoswin_oracle as po
import ref_loader
from _util import TINY, TINY_PITCH, ZERO_GRAD_KEYS
from detfill import det_fill_module, det_uniform

pytestmark = pytest.mark.skipif(not ref_loader.reference_available(), reason="reference tree not present")


def _ref_model(ns, cfg, pano, tag):
    m = ns.SimplePanoSwinTransformer(**cfg, pano_mode=pano)
    for mod in m.modules():
        if hasattr(mod, "sphere_position_alpha_table_Te"):
            mod.sphere_position_alpha_table_Te = nn.Parameter(torch.zeros_like(mod.sphere_position_alpha_table_Te))
            mod.sphere_position_beta_table_Te = nn.Parameter(torch.zeros_like(mod.sphere_position_beta_table_Te))
    det_fill_module(m, tag)
    nn.Module.train(m, True)
    return m


@pytest.mark.parametrize("cfg,pano,shape", [(TINY, True, (2, 3, 64, 128)), (TINY, False, (1, 3, 60, 100)),
                                            (TINY_PITCH, True, (2, 3, 64, 128)), (TINY, True, (1, 3, 100, 196))])
def test_same_outputs_and_grads(cfg, pano, shape):
    ns = ref_loader.load_reference()
    ref = _ref_model(ns, cfg, pano, "live")
    ora = po.SimplePanoSwinTransformerOracle(**cfg, pano_mode=pano)
    ora.load_state_dict(ref.state_dict(), strict=True)          # identical key set
    x = det_uniform(shape, "live:x", 1.0)
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    o1, o2 = ref(x1), ora(x2)
    for a, b in zip(o1, o2):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    ws = [det_uniform(tuple(o.shape), f"live:w{i}") for i, o in enumerate(o1)]
    sum((a * w).sum() for a, w in zip(o1, ws)).backward()
    sum((a * w).sum() for a, w in zip(o2, ws)).backward()
    assert torch.allclose(x1.grad, x2.grad, rtol=1e-4, atol=1e-5 * x1.grad.abs().max().item())
    g2 = dict(ora.named_parameters())
    for k, p in ref.named_parameters():
        if p.grad is None:
            assert g2[k].grad is None or float(g2[k].grad.abs().max()) == 0.0, k
            continue
        if any(z in k for z in ZERO_GRAD_KEYS):
            continue
        scale = p.grad.abs().max().item()
        assert torch.allclose(p.grad, g2[k].grad, rtol=1e-4, atol=1e-5 * max(scale, 1e-6)), k


def test_reference_registry_name_and_kwargs():
    import inspect
    ns = ref_loader.load_reference()
    assert "SimplePanoSwinTransformer" in ns.BACKBONES.module_dict
    ref_sig = inspect.signature(ns.SimplePanoSwinTransformer.__init__)
    ora_sig = inspect.signature(po.SimplePanoSwinTransformerOracle.__init__)
    assert list(ref_sig.parameters) == list(ora_sig.parameters)
    for k, p in ref_sig.parameters.items():
        if k not in ("self", "norm_layer"):
            assert p.default == ora_sig.parameters[k].default, k


def test_product_class_matches_the_reference_interface():
    """The PRODUCT class (not only the oracle): same constructor parameters and defaults as the live reference (HOT:781-801)
    plus the single keyword ``compute_dtype``, and identical state-dict keys and shapes for pano / planar / pitch-block
    configurations, so reference checkpoints load with strict=True.  Construction needs no GPU."""
    import inspect
    from panoswintransformerobjectdetection_amd import SimplePanoSwinTransformer
    ns = ref_loader.load_reference()
    ref_sig = inspect.signature(ns.SimplePanoSwinTransformer.__init__).parameters
    got_sig = inspect.signature(SimplePanoSwinTransformer.__init__).parameters
    assert list(got_sig) == list(ref_sig) + ["compute_dtype"]
    for k, p in ref_sig.items():
        if k not in ("self", "norm_layer"):
            assert p.default == got_sig[k].default, k
    for cfg in (TINY, TINY_PITCH, dict(TINY, ape=False, pano_mode=False), dict(TINY, out_indices=(1, 3)),
                dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], ape=True)):
        ref = ns.SimplePanoSwinTransformer(**cfg).state_dict()
        got = SimplePanoSwinTransformer(**cfg).state_dict()
        assert list(ref.keys()) == list(got.keys()), cfg
        for k in ref:
            assert ref[k].shape == got[k].shape and ref[k].dtype == got[k].dtype, k
    # loading a reference state dict into the product is strict-clean
    m = SimplePanoSwinTransformer(**TINY_PITCH)
    m.load_state_dict(ns.SimplePanoSwinTransformer(**TINY_PITCH).state_dict(), strict=True)
