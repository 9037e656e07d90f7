"""AdamW over the one flat parameter buffer of a model, as ONE HIP launch per step (pswin_adamw_flat).

The reference trains with torch.optim.AdamW through mmcv's OptimizerHook (mmdet/apis/train.py:91-112; configs/swin/*.py: lr 1e-4,
betas (0.9, 0.999), weight_decay 0.05).  With every parameter a view of one flat fp32 buffer (dp.GradReducer.flatten_parameters) the
update is a single streaming pass, and the same pass writes the bf16 copy of the updated weights that the next forward pass's
kernels read (backbone._refresh_lowp otherwise makes that copy at the start of every forward).  Element for element the arithmetic
of torch.optim.AdamW; the step counter lives on the device, so `step()` can be captured into a hipGraph and replayed.
"""
import torch

from ._lib import PswinError
from .ops import call, ptr


# paramwise_cfg of the reference's Swin configs (configs/swin/mask_rcnn_swin_tiny_patch4_window7_mstrain_480-800_adamw_1x_coco.py:64-67):
# no weight decay on parameters whose name contains one of these keys.  Only 'norm' occurs in this backbone.
REFERENCE_PARAMWISE_CFG = dict(custom_keys={"absolute_pos_embed": dict(decay_mult=0.), "relative_position_bias_table": dict(decay_mult=0.),
                                            "norm": dict(decay_mult=0.)})


def paramwise_groups(named_params, paramwise_cfg, prefix=""):
    """[(name, lr_mult, decay_mult)] by the rule of mmcv's DefaultOptimizerConstructor for ``custom_keys``: the keys are tried
    longest first (ties alphabetically) and the first one that is a substring of the full parameter name decides."""
    keys = (paramwise_cfg or {}).get("custom_keys", {})
    order = sorted(sorted(keys), key=len, reverse=True)
    out = []
    for name, _ in named_params:
        full = f"{prefix}.{name}" if prefix else name
        lr_mult = decay_mult = 1.0
        for k in order:
            if k in full:
                lr_mult, decay_mult = float(keys[k].get("lr_mult", 1.0)), float(keys[k].get("decay_mult", 1.0))
                break
        out.append((name, lr_mult, decay_mult))
    return out


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, flat_param, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, model=None, paramwise_cfg=None, prefix="backbone"):
        """flat_param: the nn.Parameter returned by GradReducer.flatten_parameters (its .grad is the flat gradient buffer).
        model: the module whose bf16 Linear shadows are views of one flat bf16 buffer (flatten_parameters(model, torch.bfloat16));
        they are then refreshed by this optimizer's step instead of by the next forward pass.
        paramwise_cfg: mmcv-style ``dict(custom_keys={substring: dict(lr_mult=.., decay_mult=..)})`` (REFERENCE_PARAMWISE_CFG is the
        reference configs'); needs `model` to find every parameter's slot in the flat buffer.  The groups live in a byte map over the
        buffer (one byte per 4 elements) and the update stays ONE launch; `prefix` is the name the model carries inside the detector
        ('backbone' in mmdet's two-stage detectors: the keys are matched against ``backbone.<parameter name>``)."""
        if not (isinstance(flat_param, torch.nn.Parameter) and flat_param.dim() == 1 and flat_param.dtype == torch.float32):
            raise PswinError("FlatAdamW wants the one flat fp32 parameter of GradReducer.flatten_parameters")
        if flat_param.numel() % 4 or not flat_param.is_cuda:
            raise PswinError("FlatAdamW: the flat buffer must live on the GPU and hold a multiple of 4 elements")
        super().__init__([flat_param], dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.flat = flat_param
        # the state lives where torch.optim.AdamW(capturable=True) keeps it, under the same keys: state_dict() / load_state_dict()
        # of the two optimizers are interchangeable for the flat parameter
        self.state[flat_param] = dict(step=torch.zeros((), dtype=torch.float32, device=flat_param.device),      # steps taken so far
                                      exp_avg=torch.zeros_like(flat_param.data), exp_avg_sq=torch.zeros_like(flat_param.data))
        self.lowp = None
        self._model = None
        self.group_of = None                 # uint8 [n / 4]: parameter group of every 16-byte granule, or None = one group
        self.group_mults = [(1.0, 1.0)]      # (lr_mult, decay_mult) per group; group 0 = the base group (and the alignment gaps)
        if paramwise_cfg:
            if model is None:
                raise PswinError("FlatAdamW(paramwise_cfg=...) needs `model` to locate the parameters in the flat buffer")
            self._build_groups(model, paramwise_cfg, prefix)
        pair = None if model is None else model.__dict__.get("_flat_pair")
        if pair is not None:
            if pair[0].data_ptr() != flat_param.data_ptr() or pair[1].dtype != torch.bfloat16:
                raise PswinError("FlatAdamW: the model's flat shadow does not belong to this flat parameter")
            import weakref
            self.lowp = pair[1]
            self._model = weakref.ref(model)
            model.__dict__["_lowp_external"] = weakref.ref(self)      # backbone._refresh_lowp: the shadow is kept fresh here ...
            self.sync_lowp()                                          # ... starting in step with the weights as they are now

    def _build_groups(self, model, paramwise_cfg, prefix):
        import numpy as np
        n = self.flat.numel()
        base = self.flat.data_ptr()
        params = list(model.named_parameters())
        gmap = np.zeros(n // 4, dtype=np.uint8)
        for (name, p), (_, lr_mult, decay_mult) in zip(params, paramwise_groups(params, paramwise_cfg, prefix)):
            off = (p.data_ptr() - base) // 4
            if p.data_ptr() < base or off + p.numel() > n or off % 4:
                raise PswinError(f"FlatAdamW: parameter {name} is not a 16-byte aligned view of the flat buffer")
            if (lr_mult, decay_mult) not in self.group_mults:
                self.group_mults.append((lr_mult, decay_mult))
            gmap[off // 4:(off + p.numel() + 3) // 4] = self.group_mults.index((lr_mult, decay_mult))
        if len(self.group_mults) > 8:
            raise PswinError("FlatAdamW: at most 8 distinct (lr_mult, decay_mult) pairs (PSWIN_ADAMW_MAX_GROUPS)")
        if len(self.group_mults) > 1:
            self.group_of = torch.from_numpy(gmap).to(self.flat.device)
        self.param_groups[0]["paramwise_cfg"] = paramwise_cfg

    exp_avg = property(lambda self: self.state[self.flat]["exp_avg"])
    exp_avg_sq = property(lambda self: self.state[self.flat]["exp_avg_sq"])
    step_t = property(lambda self: self.state[self.flat]["step"])

    @torch.no_grad()
    def sync_lowp(self):
        """Refresh the bf16 shadow from the master weights now.  Needed by hand only where the model cannot notice a change by
        itself: weights edited between REPLAYS of a captured step (no Python runs there), or through raw pointers / ``p.data``
        without model.mark_weights_changed().  Everything PyTorch tracks (load_state_dict, nn.init, another optimizer's step) is
        detected by the next forward pass (backbone._weights_signature)."""
        if self.lowp is not None:
            self.lowp.copy_(self.flat.data)
            m = self._model() if getattr(self, "_model", None) is not None else None
            if m is not None:
                m.__dict__["_lowp_sig"] = m._weights_signature()

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise PswinError("FlatAdamW does not take a closure")
        g = self.flat.grad
        if g is None:
            return None
        if g.dtype != torch.float32 or not g.is_contiguous() or g.numel() != self.flat.numel():
            raise PswinError("FlatAdamW: the gradient must be the flat fp32 gradient buffer")
        grp = self.param_groups[0]
        st = self.state[self.flat]
        if st["step"].dtype != torch.float32 or not st["step"].is_cuda:                   # (a state dict saved by torch's non-capturable AdamW)
            st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32, device=self.flat.device)
        st["step"] += 1.0
        if self.group_of is None:
            call("pswin_adamw_flat", self.flat, ptr(self.flat.data), ptr(g), ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(self.lowp),
                 self.flat.numel(), float(grp["lr"]), float(grp["betas"][0]), float(grp["betas"][1]), float(grp["eps"]),
                 float(grp["weight_decay"]), ptr(self.step_t),
                 algo_bytes=self.flat.numel() * (28 + (2 if self.lowp is not None else 0)))
            return None
        import ctypes
        k = len(self.group_mults)
        lr_m = (ctypes.c_float * k)(*[a for a, _ in self.group_mults])
        dc_m = (ctypes.c_float * k)(*[b for _, b in self.group_mults])
        call("pswin_adamw_flat_groups", self.flat, ptr(self.flat.data), ptr(g), ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(self.lowp),
             self.flat.numel(), ptr(self.group_of), k, ctypes.cast(lr_m, ctypes.c_void_p), ctypes.cast(dc_m, ctypes.c_void_p),
             float(grp["lr"]), float(grp["betas"][0]), float(grp["betas"][1]), float(grp["eps"]), float(grp["weight_decay"]),
             ptr(self.step_t), algo_bytes=self.flat.numel() * (28.25 + (2 if self.lowp is not None else 0)))
        return None
