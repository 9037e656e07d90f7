"""The product's checkpoint loader against the REFERENCE's (mmcv_custom/checkpoint.py:286-356), through the fixture
tests/golden/checkpoint_loader.npz that oracle/gen_golden_checkpoint.py wrote by running the reference function on the same
synthetic checkpoint files (tensors reproducible from integers: oracle/detfill.py).  CPU only."""
import logging

import numpy as np
import pytest
import torch

from _util import TINY, golden
from detfill import det_fill_module
import gen_golden_checkpoint as gg

from panoswintransformerobjectdetection_amd import SimplePanoSwinTransformer
from panoswintransformerobjectdetection_amd.checkpoint import convert_state_dict, load_checkpoint, load_state_dict

CASES = ("wrapped_module", "model_partial", "moby", "plain_partial")
# the one deliberate deviation (checkpoint.py docstring): the reference's replace('encoder.', '') mangles abs_encoder.* in MoBY files
MOBY_FIXED = ("abs_encoder.weight", "abs_encoder.bias")


class _Capture(logging.Handler):
    def __init__(self):
        super().__init__()
        self.msgs = []

    def emit(self, record):
        self.msgs.append(record.getMessage())


def _load(case, tmp_path):
    m = SimplePanoSwinTransformer(**TINY, pano_mode=True)
    det_fill_module(m, "ckpt_init")
    obj = gg.ckpt_cases(m.state_dict())[case]
    path = tmp_path / (case + ".pth")
    torch.save(obj, path)
    logger = logging.getLogger("ckpt_test_" + case)
    logger.propagate = False
    cap = _Capture()
    logger.addHandler(cap)
    ret = load_checkpoint(m, str(path), map_location="cpu", strict=False, logger=logger)
    return m, obj, ret, "\n".join(cap.msgs)


@pytest.mark.parametrize("case", CASES)
def test_loaded_state_equals_the_reference_loaders(case, tmp_path):
    g = golden("checkpoint_loader")
    m, obj, ret, report = _load(case, tmp_path)
    assert isinstance(ret, dict) and set(ret) == set(obj)                 # the reference returns the checkpoint object (:353-356)
    state = m.state_dict()
    want_keys = {k[len(case) + 4:] for k in g.files if k.startswith(case + "/fp/")}
    assert want_keys == set(state)                                        # same state-dict keys as the reference model
    for k, v in state.items():
        fp = gg.fingerprint(v).numpy()
        ref = g[f"{case}/fp/{k}"]
        if case == "moby" and k in MOBY_FIXED:
            # reference: left at the initial fill; product: the checkpoint's tensor
            assert not np.allclose(fp, ref)
            assert torch.equal(v, gg.ckpt_tensor("moby", k, v))
            continue
        np.testing.assert_allclose(fp, ref, rtol=0, atol=0, err_msg=k)
    ref_report = str(g[case + "/report"])
    if case == "moby":
        assert "abs_weight" in ref_report and report == ""              # the reference's report shows its own defect; ours is clean
    else:
        assert report == ref_report                                       # same wording, same key order


def test_table_resize_equals_the_reference(tmp_path):
    """A 5 x 5-window relative_position_bias_table (81 rows) lands in the 7 x 7 planar table with the reference's bicubic values."""
    g = golden("checkpoint_loader")
    want = torch.from_numpy(g["table_resize/table"])
    # (a) the holder module the reference was run on: same key, same numbers
    holder = gg.TableHolder()
    path = tmp_path / "table.pth"
    torch.save(gg.table_case(), path)
    load_checkpoint(holder, str(path))
    assert torch.equal(holder.state_dict()[gg.TABLE_KEY], want)
    # (b) the product backbone has no such key: the table goes to the planar (beta) table of that block
    cfg = dict(embed_dim=96, depths=[2, 2, 2, 2], num_heads=[3, 6, 12, 24], ape=True, drop_path_rate=0.0)
    m = SimplePanoSwinTransformer(**cfg)
    sd = convert_state_dict(gg.table_case(), m.state_dict())
    key = "layers.0.blocks.0.attn.sphere_position_beta_table_Te"
    assert list(sd) == [key] and torch.equal(sd[key], want)


def test_strict_raises_with_the_reference_wording(tmp_path):
    m = SimplePanoSwinTransformer(**TINY, pano_mode=True)
    sd = {k: v for k, v in m.state_dict().items() if not k.startswith("norm3")}
    sd["stray"] = torch.zeros(1)
    with pytest.raises(RuntimeError, match="The model and loaded state dict do not match exactly"):
        load_state_dict(m, sd, strict=True)
    missing, unexpected = load_state_dict(m, sd, strict=False, logger=logging.getLogger("quiet"))
    assert missing == ["norm3.weight", "norm3.bias"] and unexpected == ["stray"]
    with pytest.raises(RuntimeError, match="No state_dict found"):
        p = tmp_path / "bad.pth"
        torch.save([1, 2, 3], p)
        load_checkpoint(m, str(p))


def test_paramwise_groups_follow_the_mmcv_rule():
    """custom_keys of the reference configs (configs/swin/mask_rcnn_swin_tiny_..._1x_coco.py:64-67): substring match on the full name,
    longest key first."""
    from panoswintransformerobjectdetection_amd.optim import REFERENCE_PARAMWISE_CFG, paramwise_groups
    m = SimplePanoSwinTransformer(**TINY, pano_mode=True)
    named = list(m.named_parameters())
    got = {n: (a, b) for n, a, b in paramwise_groups(named, REFERENCE_PARAMWISE_CFG, "backbone")}
    for n, _ in named:
        assert got[n] == ((1.0, 0.0) if "norm" in n else (1.0, 1.0)), n
    assert got["patch_embed.norm.weight"] == (1.0, 0.0) and got["patch_embed.proj.1.weight"] == (1.0, 1.0)     # BatchNorm has no 'norm' in its name
    assert got["layers.0.blocks.0.attn.sphere_position_beta_table_Te"] == (1.0, 1.0)                            # no key of the config matches the tables
    pw = dict(custom_keys={"norm": dict(decay_mult=0.), "layers.0.blocks.0.norm1": dict(lr_mult=2.0)})
    got = {n: (a, b) for n, a, b in paramwise_groups(named, pw)}
    assert got["layers.0.blocks.0.norm1.weight"] == (2.0, 1.0) and got["layers.0.blocks.1.norm1.weight"] == (1.0, 0.0)


def test_loading_through_a_wrapper_marks_the_nested_backbone_weights_changed(tmp_path):
    """load_checkpoint on a detector-like wrapper writes the nested backbone's weights; its low-precision shadows are stale
    unless the BACKBONE's epoch is bumped (mark_weights_changed on every submodule that has it)."""
    class Wrapper(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = SimplePanoSwinTransformer(**TINY, pano_mode=True)

    w = Wrapper()
    path = tmp_path / "wrapped.pth"
    torch.save({"state_dict": w.state_dict()}, path)
    before = w.backbone.__dict__.get("_lowp_epoch", 0)
    load_checkpoint(w, str(path), strict=False)
    assert w.backbone.__dict__.get("_lowp_epoch", 0) == before + 1
