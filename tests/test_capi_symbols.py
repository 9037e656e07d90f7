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
