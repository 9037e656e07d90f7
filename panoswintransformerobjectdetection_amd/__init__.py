"""MI355X-native PanoSwin windowed-attention backbone (drop-in for the reference's
mmdet/models/backbones/simple_panoswin_transformer.py).  Importing this package registers
``SimplePanoSwinTransformer`` in the BACKBONES registry."""
from ._lib import PswinError, LIB_PATH  # noqa: F401
from .registry import BACKBONES, build_backbone  # noqa: F401
from .backbone import (BasicLayer, PanoSwinTransformerBlock, PatchEmbed, PatchMerging,  # noqa: F401
                       PitchAttentionModule, SimplePanoSwinTransformer, WindowAttention)
from . import ops  # noqa: F401

__all__ = ["SimplePanoSwinTransformer", "BACKBONES", "build_backbone", "ops", "PswinError"]
