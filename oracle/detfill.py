"""Deterministic, RNG-free tensor fills (TEST INFRASTRUCTURE ONLY).

Golden fixtures must be reproducible without storing megabytes of weights and without depending on a
torch RNG stream.  ``det_fill`` derives every element from a splitmix64 hash of (tag, flat index) using
integer arithmetic only, so numpy reproduces it bit for bit on any machine.
"""
import hashlib

import numpy as np
import torch

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
        return x ^ (x >> np.uint64(31))


def det_uniform(shape, tag, scale=1.0, offset=0.0):
    """float32 tensor, elements uniform-looking in offset + [-scale, scale), a pure function of (tag, index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(int.from_bytes(hashlib.sha256(tag.encode()).digest()[:8], "little"))
    with np.errstate(over="ignore"):
        h = _splitmix64((np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + seed) & _M)
    u = (h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))      # [0, 1), 24 bits
    out = (u - np.float32(0.5)) * np.float32(2.0 * scale) + np.float32(offset)
    return torch.from_numpy(out.reshape(shape).astype(np.float32))


@torch.no_grad()
def det_fill_module(module, tag="w"):
    """Fill every parameter / float buffer of a module from its state-dict key.

    Weights ~ U(-s, s) with s = 1.5 / sqrt(fan_in) (keeps activations O(1) through the depth), biases
    small, norm gains around 1, BatchNorm running_var positive, the alpha / beta bias tables distinct
    and large enough (0.3) for their gradients to be well away from rounding noise.
    """
    sd = module.state_dict()
    for k, v in sd.items():
        if not torch.is_floating_point(v):
            continue
        key = f"{tag}:{k}"
        if k.endswith("running_var"):
            new = det_uniform(v.shape, key, 0.25, 1.0)
        elif k.endswith("running_mean"):
            new = det_uniform(v.shape, key, 0.1)
        elif "table" in k:
            new = det_uniform(v.shape, key, 0.3)
        elif k.endswith("np_uv"):
            continue
        elif v.dim() >= 2:
            fan_in = int(np.prod(v.shape[1:]))
            new = det_uniform(v.shape, key, 1.5 / np.sqrt(fan_in))
        elif "norm" in k and k.endswith("weight") or (".proj.1." in k or ".proj.4." in k) and k.endswith("weight"):
            new = det_uniform(v.shape, key, 0.2, 1.0)
        else:
            new = det_uniform(v.shape, key, 0.1)
        v.copy_(new)
    module.load_state_dict(sd)
    return module
