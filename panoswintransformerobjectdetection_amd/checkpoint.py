"""Checkpoint loading for ``init_weights(pretrained=str)`` (reference: mmcv_custom/checkpoint.py:286-356).

State-dict keys are identical to the reference's, so its checkpoints load directly.  A vanilla Swin
checkpoint (which the reference's loader cannot handle: it indexes a missing key, checkpoint.py:339) is
accepted by mapping ``relative_position_bias_table`` onto the planar table ``sphere_position_beta_table_Te``.
"""
import logging

import torch
import torch.nn.functional as F


def _unwrap(ckpt):
    for key in ("state_dict", "model"):
        if isinstance(ckpt, dict) and key in ckpt and isinstance(ckpt[key], dict):
            return ckpt[key]
    return ckpt


def load_checkpoint(model, filename, map_location="cpu", strict=False, logger=None):
    logger = logger or logging.getLogger("panoswin")
    sd = _unwrap(torch.load(filename, map_location=map_location))
    if not isinstance(sd, dict):
        raise RuntimeError(f"No state_dict found in checkpoint file {filename}")          # checkpoint.py:305-307
    if sd and sorted(sd.keys())[0].startswith("encoder"):                                  # MoBY online branch, :320-321
        sd = {k.replace("encoder.", ""): v for k, v in sd.items() if k.startswith("encoder.")}
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
    # position-bias tables trained with another window size: bicubic resize of the (2w-1)^2 grid, checkpoint.py:335-351
    for k in [k for k in out if k.endswith("sphere_position_beta_table_Te") or k.endswith("sphere_position_alpha_table_Te")]:
        if k in own and out[k].dim() == 2 and out[k].shape[1] == own[k].shape[1] and out[k].shape[0] != own[k].shape[0]:
            l1, nh = out[k].shape
            l2 = own[k].shape[0]
            s1, s2 = int(l1 ** 0.5), int(l2 ** 0.5)
            if s1 * s1 == l1 and s2 * s2 == l2:
                t = F.interpolate(out[k].float().permute(1, 0).reshape(1, nh, s1, s1), size=(s2, s2), mode="bicubic")
                out[k] = t.reshape(nh, l2).permute(1, 0).contiguous()
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
