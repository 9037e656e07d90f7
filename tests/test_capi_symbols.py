"""The C-ABI library loads on a CPU-only host and exports every symbol include/pswin.h declares (no compute)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "pswin.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(pswin_\w+)\s*\(", text)))


def test_header_and_binding_agree():
    from panoswintransformerobjectdetection_amd import _lib
    assert _declared() == _lib.exported_symbols()


def test_library_exports_every_declared_symbol():
    from panoswintransformerobjectdetection_amd import _lib, build
    build.build(force=False, verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.pswin_version() == _lib.ABI_VERSION
    # host-only helpers run without a GPU
    hp, wp, nw = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.pswin_window_grid(1, 128, 256, ctypes.byref(hp), ctypes.byref(wp), ctypes.byref(nw)) == 0
    assert (hp.value, wp.value, nw.value) == (259, 133, 703)          # SURVEY section 8 geometry table, stage 0
    assert lib.pswin_window_grid(0, 128, 256, ctypes.byref(hp), ctypes.byref(wp), ctypes.byref(nw)) == 0
    assert (hp.value, wp.value, nw.value) == (133, 259, 703)
    assert lib.pswin_window_grid(7, 1, 1, None, None, None) == -1    # PSWIN_ERR_ARG
    assert lib.pswin_attn_suggest_chunks(8 * 703, 703, 3, 0) == 1 and lib.pswin_attn_suggest_chunks(8 * 15, 15, 24, 1) == 2
    assert lib.pswin_attn_table_grads_workspace(3) > 0


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "panoswintransformerobjectdetection_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "panoswin_oracle" not in src and "ref_loader" not in src and "import oracle" not in src, fn


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from panoswintransformerobjectdetection_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.PswinError):
        _lib.load()


def test_checkpoint_loader_semantics(tmp_path):
    """init_weights(pretrained=path): prefix stripping, state_dict / model wrappers, Swin key remap, bicubic resize of a
    position-bias table trained with another window size (reference: mmcv_custom/checkpoint.py:286-356)."""
    import torch
    import torch.nn.functional as F
    from panoswintransformerobjectdetection_amd import SimplePanoSwinTransformer
    from panoswintransformerobjectdetection_amd.checkpoint import load_checkpoint
    cfg = dict(embed_dim=96, depths=[2, 2, 2, 2], num_heads=[3, 6, 12, 24], window_size=7, ape=True, pano_mode=True)
    torch.manual_seed(0)
    src = SimplePanoSwinTransformer(**cfg)
    sd = {k: torch.randn_like(v) if v.is_floating_point() else v.clone() for k, v in src.state_dict().items()}
    wrapped = {"state_dict": {"module.backbone." + k: v for k, v in sd.items()}}
    wrapped["state_dict"]["module.neck.lateral.weight"] = torch.zeros(3)                     # detector parts are skipped
    # a vanilla-Swin style table for window 12 ((2*12-1)^2 = 529 rows) under the reference Swin key
    key = "layers.0.blocks.0.attn.sphere_position_beta_table_Te"
    big = torch.randn(529, 3)
    del wrapped["state_dict"]["module.backbone." + key]
    wrapped["state_dict"]["module.backbone.layers.0.blocks.0.attn.relative_position_bias_table"] = big
    path = tmp_path / "ckpt.pth"
    torch.save(wrapped, path)
    dst = SimplePanoSwinTransformer(**cfg)
    missing, unexpected = load_checkpoint(dst, str(path))
    assert not missing and not unexpected
    got = dst.state_dict()
    for k, v in sd.items():
        if k != key:
            assert torch.equal(got[k], v), k
    want = F.interpolate(big.permute(1, 0).reshape(1, 3, 23, 23), size=(13, 13), mode="bicubic").reshape(3, 169).permute(1, 0)
    assert torch.allclose(got[key], want)
    # init_weights goes through the same loader
    dst2 = SimplePanoSwinTransformer(**cfg)
    dst2.init_weights(str(path))
    assert torch.equal(dst2.state_dict()["patch_embed.proj.0.weight"], sd["patch_embed.proj.0.weight"])
