"""Host-side rules of the product that need no GPU: the feature list behind PSWIN_DISABLE, the row-tile rule of the tiled GEMM, the
row-split rule of the grouped weight gradients (the kernel's XCD-aware map expects 1, 2, 4 or a multiple of 8 splits)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ops():
    import panoswintransformerobjectdetection_amd.ops as ops
    return ops


def test_pswin_disable_turns_named_features_off_and_nothing_else():
    """(in a process of its own: the switches are read once, at import)"""
    code = ("import panoswintransformerobjectdetection_amd.ops as o; "
            "print(int(o.GROUPED_WGRAD), int(o.GEMM_NT), int(o.FUSED_WINDOW_ATTENTION), int(o.FUSED_MLP), int(o.LN_FUSED_MOVES), "
            "int(o.DEFER_TABLE_PARTIALS), int(o.GEMM_TN_RING), o.gemm_nt_tile(16384, 384, 1536))")
    env = dict(os.environ, PSWIN_DISABLE="grouped_wgrad, GEMM_NT", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["0", "0", "1", "1", "1", "1", "1", "0"]                   # gemm_nt off: the library takes the layer (tile 0)
    ops = _ops()
    if not os.environ.get("PSWIN_DISABLE"):
        assert ops.GROUPED_WGRAD and ops.GEMM_NT


@pytest.mark.parametrize("M,K,N,tile", [
    (16384, 384, 1536, 128),      # stage 2 fc1, batch 8: 1,024 tiles of 128 rows = two rounds of the 512 tile slots
    (16384, 1536, 384, 128),      # stage 2 fc2: 512 tiles of 64 rows, but 256 of 128 rows fit the chip once (four-stage form)
    (19600, 384, 384, 96),        # stage 2 proj: 614 tiles of 64 rows spill into a second round, 410 of 96 rows do not
    (19600, 1152, 384, 96),       # stage 2 qkv data gradient
    (74480, 192, 576, 128),       # stage 1 qkv: many rounds
    (5880, 768, 2304, 128),       # stage 3 qkv forward at batch 8: 552 tiles of 128 rows -> the HIP kernel
    (4096, 3072, 768, 0),         # stage 3 fc2 forward: long contraction over 128 tiles -> the library
    (4900, 384, 1152, 0),         # stage 2 qkv at batch 2 -> the library
    (4900, 384, 384, 64),         # stage 2 proj at batch 2: 154 tiles, four-stage form
])
def test_row_tile_rule_of_the_tiled_gemm(M, K, N, tile):
    assert _ops().gemm_nt_tile(M, K, N) == tile


def test_row_splits_of_the_grouped_weight_gradients_divide_over_the_xcds():
    ops = _ops()
    for M in (64, 500, 1024, 1470, 4096, 4900, 5880, 9000, 16384, 19600, 65536, 74480, 262144, 275576):
        s = ops.grouped_wgrad_splits(M)
        assert s in (1, 2, 4) or s % 8 == 0, (M, s)
        assert 1 <= s <= max(1, M // 64)
        if M >= 4096:
            assert 1024 <= M / s <= 3072, (M, s)                                # about 2,048 rows per workgroup
    assert ops.grouped_wgrad_splits(19600) == 8 and ops.grouped_wgrad_splits(5880) == 4 and ops.grouped_wgrad_splits(4096) == 2
