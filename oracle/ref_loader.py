"""Dev-container-only loader for the REAL reference hot path (test infrastructure, never shipped).

TEST INFRASTRUCTURE ONLY.  Nothing under ``panoswintransformerobjectdetection_amd/`` may import this
file.  It is used by ``oracle/gen_golden.py`` (to produce the fixtures under ``tests/golden/``) and by
the ``not gpu`` tests that re-check the CPU restatement against the live reference when
``/root/reference`` exists.  ``/root/reference`` does not exist on the GPU box, so everything here
degrades to "reference unavailable" (``load_reference()`` returns ``None``).

What it does (SURVEY.md section 8c "Loader recipe"): the reference file
``mmdet/models/backbones/simple_panoswin_transformer.py`` (HOT) imports third-party packages that are
not installed in this image (timm, mmcv, mmdet, cv2, fvcore, thop, ...).  None of them contributes
arithmetic to the path except ``timm.DropPath`` / ``trunc_normal_`` whose semantics are restated
below.  We pre-seed ``sys.modules`` with minimal stand-ins for those *third-party* modules, then
import HOT, ``lzx/models/great_circle.py`` (for real) and the pure-torch slice of
``lzx/pano_rotate.py`` *from where they lie* under ``/root/reference`` -- no reference source is copied
into this repository.

Reference quirk D3: HOT:1038 calls ``pano_rotate_image(x, np_uv=..,
with_uv=True)`` and unpacks three values, but the shipped ``lzx/pano_rotate.py:169`` accepts no
``with_uv`` and returns two values, with a batch-1 sampling grid.  The pitch module therefore cannot run
as shipped.  ``_pano_rotate_image_shim`` applies the reference function per sample (same grid for
every sample) and returns a third ``None``; goldens produced through it are tagged
"reference-as-shimmed".
"""
import importlib.util
import logging
import math
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("PSWIN_REFERENCE_ROOT", "/root/reference")
HOT_REL = "mmdet/models/backbones/simple_panoswin_transformer.py"

_CACHE = {}


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, HOT_REL))


# ----------------------------------------------------------------------------------------------
# third-party stand-ins (timm 0.3/0.4 semantics; requirements/runtime.txt lists `timm` unpinned)
# ----------------------------------------------------------------------------------------------
class _DropPath(nn.Module):
    """timm.models.layers.DropPath: per-sample stochastic depth, identity in eval / p == 0."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        rnd = keep + torch.rand(shape, dtype=x.dtype, device=x.device)
        rnd.floor_()
        return x.div(keep) * rnd


def _to_2tuple(x):
    if isinstance(x, (tuple, list)):
        return tuple(x)
    return (x, x)


def _trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


class _Registry:
    """Minimal mmcv.utils.Registry: name -> class, ``register_module()`` decorator."""

    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls

        if module is not None:
            return _reg(module)
        return _reg

    def get(self, key):
        return self.module_dict.get(key)


def _mk(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_stubs():
    timm = _mk("timm")
    timm_models = _mk("timm.models")
    timm_layers = _mk("timm.models.layers", DropPath=_DropPath, to_2tuple=_to_2tuple,
                      trunc_normal_=_trunc_normal_)
    timm.models = timm_models
    timm_models.layers = timm_layers

    mmcv = _mk("mmcv")
    mmcv_utils = _mk("mmcv.utils", Registry=_Registry)
    mmcv.utils = mmcv_utils

    _mk("mmcv_custom", load_checkpoint=lambda *a, **k: None)
    mmdet = _mk("mmdet")
    mmdet_utils = _mk("mmdet.utils", get_root_logger=lambda *a, **k: logging.getLogger("mmdet"))
    mmdet.utils = mmdet_utils
    fv = _mk("fvcore")
    fv.nn = _mk("fvcore.nn", FlopCountAnalysis=None, parameter_count_table=None)
    _mk("thop", profile=None)

    lzx = _mk("lzx")
    lzx.__path__ = [os.path.join(REFERENCE_ROOT, "lzx")]
    lzx.utils = _mk("lzx.utils", cv_show1=lambda *a, **k: None)
    lzx_models = _mk("lzx.models")
    lzx_models.__path__ = [os.path.join(REFERENCE_ROOT, "lzx", "models")]
    lzx.models = lzx_models


def _load_file_module(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _load_pano_rotate():
    """Execute only the pure-torch slice of lzx/pano_rotate.py (lines 13, 16-95, 169-187).

    The module-level imports of that file drag in cv2/seaborn/pycocotools which are absent; the four
    functions on the hot path (uv2xyz, rotate, pano_rotate, pano_rotate_image) only need torch,
    einops, math and torch.nn.functional.
    """
    import einops
    import torch.nn.functional as F

    path = os.path.join(REFERENCE_ROOT, "lzx", "pano_rotate.py")
    with open(path, "r") as f:
        lines = f.read().split("\n")
    keep = [lines[12]] + lines[15:95] + lines[168:187]
    src = "\n".join(keep)
    mod = types.ModuleType("lzx.pano_rotate")
    mod.__dict__.update(dict(torch=torch, math=math, einops=einops, F=F,
                             _pano_rotate_image_s_uvs=lambda *a, **k: None))
    exec(compile(src, path + "<hot-slice>", "exec"), mod.__dict__)
    ref_pri = mod.pano_rotate_image

    def _pano_rotate_image_shim(bcwh, np_uv, tuvwh2xyxy_boxes=None, with_uv=False):
        outs = [ref_pri(bcwh[b:b + 1], np_uv)[0] for b in range(bcwh.shape[0])]
        out = torch.cat(outs, 0)
        if with_uv:
            return out, None, None
        return out, None

    mod.pano_rotate_image_reference = ref_pri
    mod.pano_rotate_image = _pano_rotate_image_shim
    sys.modules["lzx.pano_rotate"] = mod
    return mod


def load_reference():
    """Return a namespace with the reference's hot-path modules, or None when unavailable."""
    if "ns" in _CACHE:
        return _CACHE["ns"]
    if not reference_available():
        _CACHE["ns"] = None
        return None
    _install_stubs()
    gc = _load_file_module("lzx.models.great_circle",
                           os.path.join(REFERENCE_ROOT, "lzx", "models", "great_circle.py"))
    pr = _load_pano_rotate()
    hot = _load_file_module("ref_simple_panoswin_transformer", os.path.join(REFERENCE_ROOT, HOT_REL))
    ns = types.SimpleNamespace(hot=hot, great_circle=gc, pano_rotate=pr,
                               SimplePanoSwinTransformer=hot.SimplePanoSwinTransformer,
                               BACKBONES=hot.BACKBONES)
    _CACHE["ns"] = ns
    return ns


if __name__ == "__main__":
    ns = load_reference()
    print("reference available:", ns is not None)
    if ns is not None:
        print("registry:", list(ns.BACKBONES.module_dict))
        torch.manual_seed(0)
        m = ns.SimplePanoSwinTransformer(depths=[2, 2, 6, 2], ape=True, drop_path_rate=0.0)
        m.init_weights(None)
        super(type(m), m).train(False)
        with torch.no_grad():
            outs = m(torch.randn(1, 3, 128, 256))
        print([tuple(o.shape) for o in outs])


def load_reference_checkpoint_module():
    """The reference's checkpoint loader, ``mmcv_custom/checkpoint.py``, imported in place with stand-ins for the third-party
    modules its import block names (torchvision, mmcv.fileio / parallel / utils / runner: none of them is reached when a local
    file is loaded on one process).  Dev container only; returns None when /root/reference is absent."""
    if "ckpt" in _CACHE:
        return _CACHE["ckpt"]
    path = os.path.join(REFERENCE_ROOT, "mmcv_custom", "checkpoint.py")
    if not os.path.isfile(path):
        _CACHE["ckpt"] = None
        return None
    saved = {k: sys.modules.get(k) for k in ("torchvision", "mmcv", "mmcv.fileio", "mmcv.parallel", "mmcv.utils", "mmcv.runner")}
    try:
        _mk("torchvision")
        mmcv = _mk("mmcv")
        mmcv.fileio = _mk("mmcv.fileio", FileClient=object, load=lambda *a, **k: None)
        mmcv.parallel = _mk("mmcv.parallel", is_module_wrapper=lambda m: False)
        mmcv.utils = _mk("mmcv.utils", mkdir_or_exist=lambda *a, **k: None, Registry=_Registry)
        mmcv.runner = _mk("mmcv.runner", get_dist_info=lambda: (0, 1))
        mod = _load_file_module("ref_mmcv_custom_checkpoint", path)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    _CACHE["ckpt"] = mod
    return mod
