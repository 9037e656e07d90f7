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


def test_argument_errors_of_the_round_2_entry_points_without_a_gpu():
    """Every entry point validates its arguments before it touches the device: bad calls return PSWIN_ERR_ARG (-1) on a CPU-only
    host (the reference's operators raise on such inputs; the Python layer turns the code into PswinError)."""
    from panoswintransformerobjectdetection_amd import _lib
    lib = _lib.load()
    ERR = -1
    buf = (ctypes.c_char * 4096)()
    p = ctypes.cast(buf, ctypes.c_void_p).value
    p16 = (p + 15) & ~15
    # tiled GEMM family: shape support (N a multiple of 192, K of 64, M >= 64, < 4 GB operands) and tile heights
    assert lib.pswin_gemm_nt_supported(16384, 384, 1536) == 1 and lib.pswin_gemm_nt_supported(16384, 384, 1500) == 0
    assert lib.pswin_gemm_nt_supported(32, 384, 1536) == 0 and lib.pswin_gemm_nt_supported(16384, 100, 1536) == 0
    assert lib.pswin_gemm_nt(None, p16, None, p16, 16384, 384, 1536, 0, None) == ERR                      # null operand
    assert lib.pswin_gemm_nt(p16, p16, None, p16, 16384, 384, 1536, 80, None) == ERR                      # tile height (64, 96, 128)
    assert lib.pswin_gemm_nt(p16 + 2, p16, None, p16, 16384, 384, 1536, 0, None) == ERR                   # alignment
    assert lib.pswin_gemm_nt_gelu_fwd(p16, p16, None, p16, p16, 16384, 384, 1536, 128, None) == ERR       # fc1 bias is required
    assert lib.pswin_gemm_nt_gelu_fwd(p16, p16, p16, p16, p16, 16384, 384, 1536, 0, None) == ERR          # explicit tile height
    assert lib.pswin_gemm_nt_gelu_bwd(p16, p16, p16, p16, p16, None, 16384, 384, 1536, 128, None) == ERR  # partial sums buffer
    assert lib.pswin_gemm_nt_partial_rows(16384, 128) == 128 and lib.pswin_gemm_nt_partial_rows(16385, 64) == 257
    assert lib.pswin_gemm_nt_partial_rows(16384, 100) == ERR
    assert lib.pswin_gemm_tn_ring_supported(16384, 1152, 384) == 1 and lib.pswin_gemm_tn_ring_supported(16384, 576, 200) == 0
    assert lib.pswin_gemm_tn_ring(p16, p16, p16, 1, 16384, 1152, 384, 0, None) == ERR                     # splits >= 1
    # fused window kernel: C = 96 / 3 heads / bf16 only
    assert lib.pswin_win_attn_fused_supported(96, 3, 1) == 1 and lib.pswin_win_attn_fused_supported(192, 6, 1) == 0
    assert lib.pswin_win_attn_fused_supported(96, 3, 0) == 0
    # attention backward with explicit q / k / v strides: strides must cover a window and be multiples of 8 elements
    args = [p16, p16, p16, 96, 49 * 96, 32, None, 0, None, p16, None, 0, p16, 96, p16, p16, p16, p16, 96, None, 1, 8, 8, 3, 0.1767767, 1, None]
    bad = list(args); bad[4] = 48 * 96
    assert lib.pswin_attn_bwd_ex(*bad) == ERR
    bad = list(args); bad[5] = 30
    assert lib.pswin_attn_bwd_ex(*bad) == ERR
    bad = list(args); bad[20] = 3                      # chunks must divide the images per bias window (here 1)
    assert lib.pswin_attn_bwd_ex(*bad) == ERR
    # AdamW over the flat buffer: multiple of 4 elements, aligned, sane hyper-parameters, device step counter
    ok = [p16, p16, p16, p16, None, 1024, 1e-4, 0.9, 0.999, 1e-8, 0.05, p16, None]
    for i, v in ((5, 1022), (0, None), (11, None), (7, 1.0), (8, -0.1), (6, -1e-4), (0, p16 + 4)):
        bad = list(ok); bad[i] = v
        assert lib.pswin_adamw_flat(*bad) == ERR, (i, v)


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
    from panoswintransformerobjectdetection_amd.checkpoint import convert_state_dict, load_checkpoint, load_state_dict
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
    ckpt = load_checkpoint(dst, str(path))
    assert set(ckpt) == {"state_dict"}                                    # returns the checkpoint object, as the reference does
    missing, unexpected = load_state_dict(SimplePanoSwinTransformer(**cfg), convert_state_dict(wrapped, dst.state_dict()))
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


def test_supported_predicates_send_oversize_batches_to_the_unfused_path():
    """pswin_mlp0_fwd / _bwd, pswin_gemm_skinny and pswin_fc1_gelu_* reject row operands past the 32-bit buffer range with
    PSWIN_ERR_ARG; the Python predicates must say "unsupported" for those sizes first (so the caller falls through to the next path)
    and keep saying "supported" for the benched ones.  Shape logic only -- no kernel runs."""
    from panoswintransformerobjectdetection_amd import ops
    M_bench = 8 * 128 * 256                                         # PanoSwin-T stage 0 at batch 8
    assert ops.mlp0_fused_shape_ok(M_bench, 96, 384) and ops.fc1_gelu_shape_ok(M_bench, 96, 384)
    assert ops.skinny_gemm_shape_ok(M_bench, 96, 288) and ops.skinny_gemm_shape_ok(M_bench, 384, 96)
    M_big = 171 * 128 * 256                                         # batch 171 at 512 x 1024: M * 384 * 2 >= 0xFFFFFF00
    assert M_big * 384 * 2 >= 0xFFFFFF00 > (M_big - 128 * 256) * 384 * 2
    assert not ops.mlp0_fused_shape_ok(M_big, 96, 384) and not ops.fc1_gelu_shape_ok(M_big, 96, 384)
    assert not ops.skinny_gemm_shape_ok(M_big, 384, 96) and ops.skinny_gemm_shape_ok(M_big, 96, 96)
    assert ops.mlp0_fused_shape_ok(M_big - 128 * 256, 96, 384)
    assert not ops.mlp0_fused_shape_ok(2048, 96, 384)               # tiny inputs stay on the generic path as before
    assert ops.fused_windows_addressable(8 * 703, 96) and not ops.fused_windows_addressable(1 << 26, 384)
