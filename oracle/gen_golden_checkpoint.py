"""Generate tests/golden/checkpoint_loader.npz from the LIVE reference loader (dev container only; TEST INFRASTRUCTURE).

    python oracle/gen_golden_checkpoint.py          # needs /root/reference

Pins ``panoswintransformerobjectdetection_amd/checkpoint.py`` to ``mmcv_custom/checkpoint.py:286-356`` (load_checkpoint): the
reference function is run on synthetic checkpoint files whose tensors are reproducible from integers (oracle/detfill.py); what
is stored is the reference's OUTPUT -- a fingerprint of every tensor of the target model after the load (f64 sum, sum of
|x|, first four values), the interpolated position-bias table in full, and the keys the reference reports as missing / unexpected.

Cases (``ckpt_cases`` is shared with tests/test_checkpoint_golden.py, which rebuilds the same files):
  wrapped_module  {'state_dict': {'module.' + k: v}}                           prefix strip decided on the FIRST key (:316-317)
  model_partial   {'model': {k: v}} with two keys dropped and one foreign key   'model' wrapper (:311-312), non-strict report
  moby            {'model': {'encoder.' + k, 'encoder_k.' + k, 'queue'}}        MoBY online branch (:320-321)
  plain_partial   {k: v} for the first stage only                               bare state dict (:313-314)
  table_resize    a 5 x 5-window table (81 rows) into a 7 x 7 model (169 rows)  bicubic resize (:335-351); the reference indexes
                  model.state_dict()[table_key], so the target is a holder module that owns that key (a PanoSwin model does not:
                  SURVEY D14); the product maps the key onto its planar table and must produce the same numbers
"""
import logging
import os
import sys
import tempfile

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402
from detfill import det_fill_module, det_uniform  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "checkpoint_loader.npz")
TINY = dict(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], ape=True, drop_path_rate=0.0)
TABLE_KEY = "layers.0.blocks.0.attn.relative_position_bias_table"


class TableHolder(nn.Module):
    """Owns one parameter under the vanilla-Swin table key (what the reference's resize branch looks up)."""

    def __init__(self, rows=169, heads=3):
        super().__init__()
        holder = self
        for part in TABLE_KEY.split(".")[:-1]:
            child = nn.Module()
            holder.add_module(part, child)
            holder = child
        holder.register_parameter("relative_position_bias_table", nn.Parameter(torch.zeros(rows, heads)))


def ckpt_tensor(case, key, ref):
    if not ref.is_floating_point():
        return ref.clone()
    return det_uniform(tuple(ref.shape), f"ckpt:{case}:{key}", 1.0).to(ref.dtype)


def ckpt_cases(state):
    """{case: checkpoint object} for a model whose state dict is `state` (names -> tensors; only shapes / dtypes are used)."""
    keys = list(state.keys())
    cases = {}
    cases["wrapped_module"] = {"state_dict": {"module." + k: ckpt_tensor("wrapped_module", k, v) for k, v in state.items()}, "meta": {"epoch": 3}}
    drop = {keys[5], keys[-3]}
    mp = {k: ckpt_tensor("model_partial", k, v) for k, v in state.items() if k not in drop}
    mp["head.fc.weight"] = det_uniform((4, 4), "ckpt:model_partial:foreign", 1.0)
    cases["model_partial"] = {"model": mp}
    moby = {}
    for k, v in state.items():
        moby["encoder." + k] = ckpt_tensor("moby", k, v)
        moby["encoder_k." + k] = ckpt_tensor("moby_k", k, v)
    moby["queue"] = det_uniform((8, 8), "ckpt:moby:queue", 1.0)
    cases["moby"] = {"model": moby}
    cases["plain_partial"] = {k: ckpt_tensor("plain_partial", k, v) for k, v in state.items() if k.startswith(("layers.0.", "patch_embed."))}
    return cases


def table_case(heads=3):
    return {"state_dict": {TABLE_KEY: det_uniform((81, heads), "ckpt:table_resize", 1.0)}}


def fingerprint(t):
    f = t.detach().double().reshape(-1)
    head = torch.zeros(4, dtype=torch.float64)
    head[:min(4, f.numel())] = f[:4]
    return torch.cat([torch.stack([f.sum(), f.abs().sum()]), head])


class _Capture(logging.Handler):
    def __init__(self):
        super().__init__()
        self.msgs = []

    def emit(self, record):
        self.msgs.append(record.getMessage())


def main():
    ns = ref_loader.load_reference()
    ck = ref_loader.load_reference_checkpoint_module()
    assert ns is not None and ck is not None, "needs /root/reference"
    out = {}
    proto = ns.SimplePanoSwinTransformer(**TINY, pano_mode=True).state_dict()
    with tempfile.TemporaryDirectory() as d:
        for case, obj in ckpt_cases(proto).items():
            path = os.path.join(d, case + ".pth")
            torch.save(obj, path)
            m = ns.SimplePanoSwinTransformer(**TINY, pano_mode=True)
            for mod in m.modules():                      # D5: alpha / beta alias one tensor on CPU; separate them as .cuda() does
                if hasattr(mod, "sphere_position_alpha_table_Te"):
                    mod.sphere_position_alpha_table_Te = nn.Parameter(torch.zeros_like(mod.sphere_position_alpha_table_Te))
                    mod.sphere_position_beta_table_Te = nn.Parameter(torch.zeros_like(mod.sphere_position_beta_table_Te))
            det_fill_module(m, "ckpt_init")
            logger = logging.getLogger("ref_ckpt_" + case)
            logger.propagate = False
            cap = _Capture()
            logger.addHandler(cap)
            ret = ck.load_checkpoint(m, path, map_location="cpu", strict=False, logger=logger)
            assert isinstance(ret, dict)
            for k, v in m.state_dict().items():
                out[f"{case}/fp/{k}"] = fingerprint(v)
            report = "\n".join(cap.msgs)
            out[f"{case}/report"] = np.array(report)
        path = os.path.join(d, "table.pth")
        torch.save(table_case(), path)
        holder = TableHolder()
        ck.load_checkpoint(holder, path, map_location="cpu", strict=False, logger=logging.getLogger("ref_ckpt_table"))
        out["table_resize/table"] = holder.state_dict()[TABLE_KEY].detach().clone()
    np.savez_compressed(OUT, **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in out.items()})
    print(f"checkpoint_loader.npz  {os.path.getsize(OUT) / 1024:.1f} KiB, {len(out)} arrays")


if __name__ == "__main__":
    main()
