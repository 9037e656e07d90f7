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


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, flat_param, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, model=None):
        """flat_param: the nn.Parameter returned by GradReducer.flatten_parameters (its .grad is the flat gradient buffer).
        model: the module whose bf16 Linear shadows are views of one flat bf16 buffer (flatten_parameters(model, torch.bfloat16));
        they are then refreshed by this optimizer's step instead of by the next forward pass."""
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
        pair = None if model is None else model.__dict__.get("_flat_pair")
        if pair is not None:
            if pair[0].data_ptr() != flat_param.data_ptr() or pair[1].dtype != torch.bfloat16:
                raise PswinError("FlatAdamW: the model's flat shadow does not belong to this flat parameter")
            import weakref
            self.lowp = pair[1]
            self._model = weakref.ref(model)
            model.__dict__["_lowp_external"] = weakref.ref(self)      # backbone._refresh_lowp: the shadow is kept fresh here ...
            self.sync_lowp()                                          # ... starting in step with the weights as they are now

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
        call("pswin_adamw_flat", self.flat, ptr(self.flat.data), ptr(g), ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(self.lowp),
             self.flat.numel(), float(grp["lr"]), float(grp["betas"][0]), float(grp["betas"][1]), float(grp["eps"]),
             float(grp["weight_decay"]), ptr(self.step_t),
             algo_bytes=self.flat.numel() * (28 + (2 if self.lowp is not None else 0)))
        return None
