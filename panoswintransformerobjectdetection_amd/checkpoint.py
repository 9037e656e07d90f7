"""Checkpoint loading for ``init_weights(pretrained=str)`` (reference: mmcv_custom/checkpoint.py:40-100 ``load_state_dict`` and
:286-356 ``load_checkpoint``; pinned by tests/golden/checkpoint_loader.npz, which oracle/gen_golden_checkpoint.py produced by
running the reference's own function on synthetic checkpoint files).

Same call and return as the reference: ``load_checkpoint(model, filename, map_location, strict, logger) -> checkpoint``; the
mismatch report (same wording) goes to the logger, or raises RuntimeError with ``strict=True``.  State-dict keys are identical to
the reference's, so its checkpoints load directly.  Where this loader does MORE than the reference, on purpose:

  * ``module.`` is stripped per key (the reference decides on the first key and then cuts 7 characters off every key, :316-317);
  * detector checkpoints: ``backbone.`` is stripped and ``neck. / rpn_head. / roi_head. / bbox_head.`` entries are skipped;
  * MoBY checkpoints (:320-321): only the LEADING ``encoder.`` is removed.  The reference's ``k.replace('encoder.', '')`` also
    rewrites ``abs_encoder.weight`` into ``abs_weight``, so it silently leaves the absolute position encoder at its initial values
    (visible in the fixture's report: "unexpected key ... abs_weight, abs_bias"); here those two tensors are loaded;
  * a vanilla Swin checkpoint, which the reference cannot load into this model at all (it indexes
    ``model.state_dict()['...relative_position_bias_table']``, a KeyError: SURVEY D14), is accepted by mapping that key onto the
    planar table ``sphere_position_beta_table_Te`` -- resized exactly as :335-351 resizes it when the window size differs.
"""
import logging

import torch
import torch.nn.functional as F

_DETECTOR_PARTS = ("neck.", "rpn_head.", "roi_head.", "bbox_head.", "mask_head.")


def load_state_dict(module, state_dict, strict=False, logger=None):
    """mmcv_custom/checkpoint.py:40-100: non-strict ``nn.Module.load_state_dict`` whose mismatch report is always shown.  Tensors
    of the wrong shape are skipped and reported.  Returns (missing_keys, unexpected_keys) (the reference returns None)."""
    own = module.state_dict()
    err_msg = []
    usable = {}
    for k, v in state_dict.items():
        if k in own and tuple(own[k].shape) != tuple(v.shape):
            err_msg.append(f"size mismatch for {k}: copying a param with shape {tuple(v.shape)} from checkpoint, the shape in "
                           f"current model is {tuple(own[k].shape)}.")
        else:
            usable[k] = v
    res = module.load_state_dict(usable, strict=False)
    skipped = set(state_dict) - set(usable)
    missing = [k for k in res.missing_keys if "num_batches_tracked" not in k and k not in skipped]      # :87-90
    unexpected = list(res.unexpected_keys)
    if unexpected:
        err_msg.append("unexpected key in source state_dict: " + ", ".join(unexpected) + "\n")
    if missing:
        err_msg.append("missing keys in source state_dict: " + ", ".join(missing) + "\n")
    if err_msg:
        err_msg.insert(0, "The model and loaded state dict do not match exactly\n")
        msg = "\n".join(err_msg)
        if strict:
            raise RuntimeError(msg)
        (logger or logging.getLogger("panoswin")).warning(msg)
    return missing, unexpected


def convert_state_dict(checkpoint, model_state):
    """The key / tensor rewriting of load_checkpoint (:309-351) as a pure function: checkpoint object -> state dict for a model
    whose own state dict is `model_state`."""
    if "state_dict" in checkpoint and isinstance(checkpoint["state_dict"], dict):           # :309-314
        sd = checkpoint["state_dict"]
    elif "model" in checkpoint and isinstance(checkpoint["model"], dict):
        sd = checkpoint["model"]
    else:
        sd = checkpoint
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}           # :316-317
    if sd and sorted(sd.keys())[0].startswith("encoder"):                                            # MoBY online branch, :320-321
        sd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    out = {}
    for k, v in sd.items():
        if k.startswith("backbone."):
            k = k[len("backbone."):]
        elif k.startswith(_DETECTOR_PARTS):
            continue
        if k.endswith("relative_position_bias_table") and k not in model_state:
            k = k[:-len("relative_position_bias_table")] + "sphere_position_beta_table_Te"
        if k.endswith("relative_position_index") and k not in model_state:
            k = k[:-len("relative_position_index")] + "relative_position_index_OO"
        out[k] = v
    # position-bias tables trained with another window size: bicubic resize of the (2w-1)^2 grid, :335-351
    for k in [k for k in out if k.endswith(("relative_position_bias_table", "sphere_position_beta_table_Te", "sphere_position_alpha_table_Te"))]:
        cur = model_state.get(k)
        if cur is None or out[k].dim() != 2 or cur.dim() != 2:
            continue
        (l1, nh1), (l2, nh2) = out[k].shape, cur.shape
        if nh1 != nh2 or l1 == l2:
            continue                                        # head mismatch: left to the size-mismatch report, as the reference (:342-343)
        s1, s2 = int(l1 ** 0.5), int(l2 ** 0.5)
        t = F.interpolate(out[k].permute(1, 0).reshape(1, nh1, s1, s1), size=(s2, s2), mode="bicubic")
        out[k] = t.reshape(nh2, l2).permute(1, 0)
    return out


def load_checkpoint(model, filename, map_location="cpu", strict=False, logger=None):
    logger = logger or logging.getLogger("panoswin")
    checkpoint = torch.load(filename, map_location=map_location)
    if not isinstance(checkpoint, dict):
        raise RuntimeError(f"No state_dict found in checkpoint file {filename}")          # :305-307
    load_state_dict(model, convert_state_dict(checkpoint, model.state_dict()), strict, logger)
    for m in model.modules():                        # the backbone may sit inside a wrapper (a detector): tell every holder of
        if hasattr(m, "mark_weights_changed"):       # low-precision weight shadows that the masters changed behind autograd's back
            m.mark_weights_changed()
    return checkpoint
