"""Checkpoint loading for ``init_weights(pretrained=str)`` (reference: mmcv_custom/checkpoint.py:286-356).

State-dict keys are identical to the reference's, so its checkpoints load directly.  A vanilla Swin
checkpoint (which the reference's loader cannot handle: it indexes a missing key, checkpoint.py:339) is
accepted by mapping ``relative_position_bias_table`` onto the planar table ``sphere_position_beta_table_Te``.
"""
import logging

import torch


def _unwrap(ckpt):
    for key in ("state_dict", "model"):
        if isinstance(ckpt, dict) and key in ckpt and isinstance(ckpt[key], dict):
            return ckpt[key]
    return ckpt


def load_checkpoint(model, filename, map_location="cpu", strict=False, logger=None):
    logger = logger or logging.getLogger("panoswin")
    sd = _unwrap(torch.load(filename, map_location=map_location))
    out = {}
    for k, v in sd.items():
        if k.startswith("module."):
            k = k[len("module."):]
        if k.startswith("backbone."):
            k = k[len("backbone."):]
        elif any(k.startswith(p) for p in ("neck.", "rpn_head.", "roi_head.", "bbox_head.")):
            continue
        if k.endswith("relative_position_bias_table"):
            k = k.replace("relative_position_bias_table", "sphere_position_beta_table_Te")
        if k.endswith("relative_position_index"):
            k = k.replace("relative_position_index", "relative_position_index_OO")
        out[k] = v
    own = model.state_dict()
    for k in list(out):
        if k in own and tuple(own[k].shape) != tuple(out[k].shape):
            logger.warning("size mismatch for %s: %s vs %s, skipped", k, tuple(out[k].shape), tuple(own[k].shape))
            del out[k]
    missing, unexpected = model.load_state_dict(out, strict=strict)
    if missing:
        logger.warning("missing keys: %s", missing)
    if unexpected:
        logger.warning("unexpected keys: %s", unexpected)
    return missing, unexpected
