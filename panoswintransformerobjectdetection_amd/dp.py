"""Data-parallel gradient averaging over RCCL (xGMI), one process per GPU.

Replaces the reference's DP wiring -- tools/dist_train.sh:8-9 (one process per device), tools/train.py:126-134
(init_dist, backend nccl) and mmdet/apis/train.py:91-99 (MMDistributedDataParallel: bucketed NCCL all-reduce of
every gradient, broadcast_buffers=False so BatchNorm statistics stay per rank, find_unused_parameters).
Contract reproduced: after ``finish()`` every rank holds mean_over_ranks(local grad) for every parameter; buffers
are never synchronised; parameters that received no gradient contribute zeros.

MI355X-first choices
  * all gradients live in ONE flat fp32 buffer (``param.grad`` are views into it): zeroing is one memset, the
    optimizer can run over contiguous memory, and a bucket is just a slice -- no flatten / unflatten copies;
  * buckets are cut in the order backward finishes the gradients -- the module's ``grad_groups()`` when it defines one
    (the backbone does: [stages 3-2 with their output norms | stages 1-0, position encoder, stem]), else reverse
    registration order -- with a forced cut at every group boundary, so a bucket never mixes late and early layers;
    each is launched as soon as its last gradient has been accumulated (post-accumulate hook) or, for the two-graph
    step, as soon as its group's backward piece has been packed, so RCCL runs under the remaining backward kernels;
  * xGMI is a point-to-point mesh (7 links per GPU): few large messages beat many small ones, hence the default
    32 MiB buckets (4 launches for the 110 MB of PanoSwin-T gradients) and ``ReduceOp.AVG`` in the collective
    itself instead of a separate divide pass.
"""
import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Rendezvous from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"     # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def completion_groups(module, groups=None):
    """Trainable parameters of `module` as a list of groups in the order a backward pass finishes them.  groups: explicit
    list of parameter lists; None: ``module.grad_groups()`` if defined, else one group in reverse registration order."""
    if groups is None and hasattr(module, "grad_groups"):
        groups = module.grad_groups()
    params = [p for p in module.parameters() if p.requires_grad]
    if groups is None:
        return [list(reversed(params))]
    groups = [[p for p in g if p.requires_grad] for g in groups]
    seen = [id(p) for g in groups for p in g]
    if len(set(seen)) != len(seen) or set(seen) != {id(p) for p in params}:
        raise ValueError("grad groups must partition the module's trainable parameters")
    return [g for g in groups if g]


class GradReducer:
    def __init__(self, module, bucket_mb=32.0, process_group=None, pack=False, groups=None):
        """pack=False: ``param.grad`` are views of the flat buffer and autograd accumulates into them (one small add
        kernel per parameter).  pack=True: gradients are produced as free tensors (``param.grad = None`` before
        backward) and gathered into the flat buffer by ONE multi-tensor copy in finish() -- fewer, larger launches;
        meant for the hipGraph path where collectives run after backward anyway.
        groups: see completion_groups; bucket boundaries are forced at group boundaries."""
        self.pack = pack
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.groups = completion_groups(module, groups)
        self.params = [p for g in self.groups for p in g]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        order = self.params                                       # the order in which backward finishes them
        group_end, n = set(), 0
        for g in self.groups:
            n += len(g)
            group_end.add(n - 1)
        offsets, total = [], 0
        for p in order:
            offsets.append(total)
            total += (p.numel() + 63) // 64 * 64                  # keep every view 256-byte aligned
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.buckets = []                                         # [start, end, n_params]
        self._bucket_of = {}
        limit = int(bucket_mb * 1024 * 1024 / 4)
        start = 0
        count = 0
        self._order, self._views = order, []
        for i, p in enumerate(order):
            self._views.append(self.flat[offsets[i]:offsets[i] + p.numel()].view_as(p))
            p.grad = None if pack else self._views[-1]
            if pack:
                # backward kernels that produce a whole parameter gradient in one piece may write it here directly
                # (ops.grad_slot); pack_grads then has nothing to copy for that parameter
                p._grad_slot = self.flat[offsets[i]:offsets[i] + p.numel()]
            self._bucket_of[p] = len(self.buckets)
            count += 1
            end = offsets[i + 1] if i + 1 < len(order) else total
            if end - start >= limit or i in group_end:
                self.buckets.append([start, end, count])
                start, count = end, 0
        self._index = {id(p): i for i, p in enumerate(order)}
        self._pending = [b[2] for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._handles = []
        self.overlap = True          # launch buckets from backward hooks; set False to reduce everything in finish()
        self._use_avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self.overlap = not pack
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params] if not pack else []

    # -- per step -----------------------------------------------------------------------------------------------
    def zero_grad(self):
        if self.pack:
            for p in self._order:
                p.grad = None
        else:
            self.flat.zero_()

    def pack_grads(self, params=None):
        """pack mode: gather the freshly produced gradients into the flat buffer and re-point ``param.grad`` at it.
        params: restrict to these parameters (a step whose backward pass runs in pieces packs each piece's share)."""
        pairs = list(zip(self._views, self._order))
        if params is not None:
            ids = {id(p) for p in params}
            pairs = [(v, p) for v, p in pairs if id(p) in ids]
        have = [(v, p.grad) for v, p in pairs if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        for v, p in pairs:
            if p.grad is None:
                v.zero_()                                         # parameter unused this step: contributes zeros
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        for v, p in pairs:
            p.grad = v

    def buckets_within(self, params):
        """Indices of the buckets that hold gradients of `params` only (they can be reduced as soon as those are packed)."""
        ids = {id(p) for p in params}
        ok, at = [], 0
        for b, (_, _, n) in enumerate(self.buckets):
            if all(id(p) in ids for p in self._order[at:at + n]):
                ok.append(b)
            at += n
        return ok

    def launch(self, bucket_ids):
        """Start the all-reduce of these buckets now (asynchronously); finish() handles the rest and waits for all."""
        if self.world > 1:
            for b in bucket_ids:
                self._launch(b)

    def _launch(self, b):
        if self._launched[b]:
            return
        self._launched[b] = True
        s, e, _ = self.buckets[b]
        view = self.flat[s:e]
        if self._use_avg:
            self._handles.append((dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group, async_op=True), None))
        else:                                                     # gloo (CPU tests): SUM then divide
            self._handles.append((dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True), view))

    def _realias(self, p, view):
        """hook mode keeps ``param.grad`` a view of the flat buffer; ``optimizer.zero_grad(set_to_none=True)`` (the torch 2
        default, and what mmcv's OptimizerHook calls every iteration) drops that view and autograd then allocates a
        fresh gradient: move it into the buffer and re-point ``param.grad`` so the collective sees it."""
        g = p.grad
        if g is None:
            view.zero_()
        elif g.data_ptr() != view.data_ptr():
            view.copy_(g)
        p.grad = view

    def _on_grad(self, p):
        if not self.pack:
            self._realias(p, self._views[self._index[id(p)]])
        if not self.overlap:
            return
        b = self._bucket_of[p]
        self._pending[b] -= 1
        if self._pending[b] == 0 and self.world > 1:
            self._launch(b)

    def finish(self):
        """Call after backward (and after pack_grads() in pack mode): launches the buckets not launched from hooks,
        waits for all of them."""
        if not self.pack:                 # parameters that got no gradient this step (or whose .grad was reset to None)
            for p, v in zip(self._order, self._views):
                if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                    if self._launched[self._bucket_of[p]]:
                        raise RuntimeError("a gradient changed after its bucket was reduced")
                    self._realias(p, v)
        if self.world > 1:
            for b in range(len(self.buckets)):
                self._launch(b)
            for h, view in self._handles:
                h.wait()
                if view is not None:
                    view.div_(self.world)
        self._handles = []
        self._pending = [b[2] for b in self.buckets]
        self._launched = [False] * len(self.buckets)

    # -- once ---------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def flatten_parameters(self, module=None, lowp_dtype=None):
        """Move every parameter into ONE flat fp32 buffer laid out like the gradient buffer (``param.data`` become views
        of it) and return it as a single nn.Parameter whose ``.grad`` is the flat gradient buffer.  An optimizer
        built on ``[flat]`` then updates the whole model with one element-wise kernel over contiguous memory instead
        of a multi-tensor launch chain over ~200 tensors (same arithmetic per element; the 64-element alignment gaps
        have zero gradient and stay zero).  Use with a single parameter group."""
        flat = torch.zeros_like(self.flat)
        for p, v in zip(self._order, self._views):
            off = v.storage_offset()
            dst = flat[off:off + p.numel()].view_as(p)
            dst.copy_(p.data)
            p.data = dst
        self.flat_param = torch.nn.Parameter(flat, requires_grad=True)
        self.flat_param.grad = self.flat
        if module is not None and lowp_dtype is not None and hasattr(module, "_attach_flat_lowp"):
            # the per-step low-precision copies of the Linear weights become views of ONE flat buffer: one cast kernel
            flat_lp = torch.empty_like(flat, dtype=lowp_dtype)
            module._attach_flat_lowp(flat, flat_lp)
        return self.flat_param

    @torch.no_grad()
    def broadcast_parameters(self, module, src=0):
        """Make every rank start from rank `src`'s weights (DDP does this at construction); buffers are left alone
        (broadcast_buffers=False in the reference)."""
        if self.world == 1:
            return
        for p in module.parameters():
            dist.broadcast(p.data, src=src, group=self.group)
        for m in module.modules():                           # p.data writes are invisible to the version counters; the backbone may be
            if hasattr(m, "mark_weights_changed"):           # nested in a wrapper (MiniMaskRCNN): every shadow holder is told
                m.mark_weights_changed()


# -- backward in two pieces (all-reduce overlapped with the second piece) ------------------------------------------
class BoundaryTap:
    """Remembers the input activation of `module` of the latest forward pass: the cut between the 'late' layers (whose
    gradients a backward pass finishes first) and the 'early' ones."""

    def __init__(self, module):
        self.x = None
        self._hook = module.register_forward_pre_hook(lambda m, a: setattr(self, "x", a[0]))

    def remove(self):
        self._hook.remove()


def split_parameters(model, late_prefixes=None):
    """(late, early) parameter lists: by name prefixes, or (None) the first group of ``model.grad_groups()`` vs the rest."""
    if late_prefixes is None:
        groups = model.grad_groups()
        return list(groups[0]), [p for g in groups[1:] for p in g]
    late = [p for n, p in model.named_parameters() if n.startswith(tuple(late_prefixes))]
    early = [p for n, p in model.named_parameters() if not n.startswith(tuple(late_prefixes))]
    return late, early


def backward_late(loss_late, xb, late_params):
    """Backward of the part of the loss that depends on the layers after the cut: sets ``.grad`` of `late_params`,
    returns d loss_late / d xb.  Together with backward_early this equals ``(loss_early + loss_late).backward()``."""
    grads = torch.autograd.grad(loss_late, [xb] + list(late_params), allow_unused=True)
    for p, g in zip(late_params, grads[1:]):
        p.grad = g
    return grads[0]


def backward_early(loss_early, xb, g_xb, early_params):
    grads = torch.autograd.grad([loss_early, xb], list(early_params),
                                grad_outputs=[torch.ones_like(loss_early), g_xb], allow_unused=True)
    for p, g in zip(early_params, grads):
        p.grad = g


def reduce_loss_scalars(losses, group=None):
    """The logged losses of one iteration averaged over the ranks with ONE small all-reduce.  The reference's
    BaseDetector._parse_losses (mmdet/models/detectors/base.py:213-218) issues one all-reduce per scalar (6-10 launches per
    iteration, each a latency-bound collective); same result: ``{name: mean over ranks of the local value}`` in sorted-key order.
    A no-op copy on one rank."""
    keys = sorted(losses)
    vals = torch.stack([losses[k].detach().float().reshape(()) for k in keys])
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(vals, op=dist.ReduceOp.SUM, group=group)
        vals = vals / dist.get_world_size(group)
    return dict(zip(keys, vals.unbind()))
