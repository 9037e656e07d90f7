"""Operators of the PanoSwin hot path: torch.autograd wrappers over the C-ABI kernels (include/pswin.h).

Every function here launches hand-written gfx950 kernels from libpswin_hip.so on the current HIP stream.
There is no PyTorch/CPU fallback: a CPU tensor or a missing library raises ``PswinError``.
Reference file:line citations use HOT = mmdet/models/backbones/simple_panoswin_transformer.py.
"""
import math
import os

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import BF16, F32, MODE_PANO, MODE_PLANAR, WPAD, WTOK, PswinError, call, dtype_code, ptr

_CACHE = {}
# Feature switches.  Every one of them selects between two HIP paths (a fused kernel or the chain of kernels it replaces), never a CPU
# path; tests flip the module attributes to compare the two.  ONE environment variable remains for same-box A/B runs of the whole step:
# PSWIN_DISABLE="name[,name...]" turns the named features off (names = the lower-case attribute names below).
_DISABLED = {n.strip().lower() for n in os.environ.get("PSWIN_DISABLE", "").split(",") if n.strip()}


def _on(name):
    return name not in _DISABLED


# the per-window qkv -> attention -> proj kernel where it exists (C = 96); off: the unfused chain
FUSED_WINDOW_ATTENTION = _on("fused_window_attention")
# the tiled HIP GEMM (pswin_gemm_nt) for the Linear layers of stages 1-3 where it measured faster than the library kernels
# (profiles/r02_gemm_nt_vs_library.txt, profiles/r04_ab_runs.json); off: library GEMMs for those layers
GEMM_NT = _on("gemm_nt")
# fc2's data gradient + the backward of fc1's bias + GELU in one kernel (pswin_gemm_nt_gelu_bwd); off: two kernels
FUSED_GELU_BWD = _on("fused_gelu_bwd")
# fc1 with the bias + GELU in its epilogue (pswin_gemm_nt_gelu_fwd) and fc2 as one autograd node; off: GEMM, then a streaming bias + GELU pass
FUSED_MLP = _on("fused_mlp")


def _dev_key(device):
    device = torch.device(device)
    return (device.type, device.index if device.index is not None else torch.cuda.current_device())


def clear_caches():
    _CACHE.clear()


# ------------------------------------------------------------------------------------------------
# parameter-gradient reductions: every "sum these partial rows" of the backward pass goes through sum_rows
# ------------------------------------------------------------------------------------------------
class _ReduceQueue:
    enabled = False
    # autograd graph task id -> jobs queued by that backward pass.  Keyed by task because passes nest (the backward of a
    # torch.utils.checkpoint segment is a pass of its own inside the outer one); each pass flushes its own jobs.
    #   "jobs": (src tensor, byte offset, dtype code, rows, cols, ld, dst tensor)
    #   "wgrad_jobs": weight gradients held back for the grouped launch (dy, x, partial, bias partial or None, M, N, K, splits, zero_lo, zero_hi)
    #   "table_jobs": attention table gradients waiting for their binning launch (after the reductions)
    #   "owners": data_ptr of every parameter that already has a postponed gradient in this pass
    tasks = {}
    slots = {}           # graph task id -> flat-gradient slots already handed out in that pass (grad_slot)

    @staticmethod
    def trim(d):         # leftovers of passes that raised: keep the table small (task ids grow monotonically)
        while len(d) > 8:
            del d[min(d)]


def _graph_task_id():
    """id of the running backward pass (-1 outside one).  A private torch entry point (present in torch 2.1 .. 2.10): without it
    nothing is postponed and no gradient slot is handed out -- every reduction launches where it is issued."""
    f = getattr(torch._C, "_current_graph_task_id", None)
    return f() if f is not None else -1


def deferred_reductions_available():
    """The two private torch entry points the end-of-pass grouping relies on (the id of the running backward pass and the
    autograd engine's end-of-pass callback) exist in this torch build."""
    eng = getattr(torch.autograd.Variable, "_execution_engine", None)
    return hasattr(torch._C, "_current_graph_task_id") and eng is not None and hasattr(eng, "queue_callback")


def set_deferred_reductions(on):
    """on=True: reductions whose result is a PARAMETER gradient (split-K partials of dW, bias-gradient partial rows,
    LayerNorm dgamma / dbeta rows) are queued while autograd runs and issued as one grouped launch when the backward pass
    ends (an autograd engine callback), instead of ~120 launches of 5-8 us each.  A postponed gradient tensor is filled
    only once backward() returns, so a reduction is postponed only when nothing can read its result earlier: the
    call site names the parameter(s) the result belongs to (``owners``), and the launch stays immediate when an owner
    already holds a ``.grad`` (autograd would accumulate into it at once), carries a tensor / post-accumulate hook
    (dp.GradReducer(pack=False) launches its all-reduce from one), or already received a postponed gradient in the same
    pass (a module applied twice: autograd adds the two as soon as the second arrives; the queue is flushed first).
    NOT covered: hooks registered on a parameter's AccumulateGrad NODE (torch DistributedDataParallel's reducer,
    ``grad_fn.register_hook`` consumers) are invisible from Python and would read unfilled gradients -- use dp.GradReducer (which
    this mode is built for) or leave deferral off under DDP.  Returns the previous setting."""
    if on and not deferred_reductions_available():
        raise PswinError("set_deferred_reductions(True) needs torch._C._current_graph_task_id and the autograd engine's queue_callback "
                         f"(private entry points, present in torch 2.1 - 2.10; this is torch {torch.__version__})")
    prev, _ReduceQueue.enabled = _ReduceQueue.enabled, bool(on)
    return prev


def _launch_reductions(jobs):
    import ctypes
    by_dev = {}
    for j in jobs:
        by_dev.setdefault(j[0].device, []).append(j)
    for lst in by_dev.values():
        arr = (_lib.ReduceJob * len(lst))()
        for a, (src, off, dt, rows, cols, ld, dst) in zip(arr, lst):
            a.src, a.dst, a.dtype, a.rows, a.cols, a.ld = src.data_ptr() + off, dst.data_ptr(), dt, rows, cols, ld
        call("pswin_reduce_jobs", lst[0][0], ctypes.cast(arr, ctypes.c_void_p), len(lst))


# off: the dScore-tile sums of an attention module run inside its backward instead of in one launch at the end of the pass
DEFER_TABLE_PARTIALS = _on("defer_table_partials")


def _launch_table_grads(jobs, stages=3):
    import ctypes
    by_dev = {}
    for j in jobs:
        by_dev.setdefault(j[0].device, []).append(j)
    for lst in by_dev.values():
        arr = (_lib.TableGradJob * len(lst))()
        for a, (gsum, dist_t, dalpha, dbeta, ws, n_tiles, nb, n_dist, heads) in zip(arr, lst):
            a.dscore_sum, a.dist_tiles_t = gsum.data_ptr(), (None if dist_t is None else dist_t.data_ptr())
            a.dalpha, a.dbeta, a.workspace = (None if dalpha is None else dalpha.data_ptr()), dbeta.data_ptr(), ws.data_ptr()
            a.n_tiles, a.n_bias_windows, a.n_dist, a.heads = n_tiles, nb, n_dist, heads
        call("pswin_attn_table_grads_batch", lst[0][0], ctypes.cast(arr, ctypes.c_void_p), len(lst), stages)


def _launch_wgrads(jobs):
    """The queued weight gradients as ONE pswin_gemm_tn_ring_jobs call per device (one kernel launch per tile geometry): longest row
    ranges first, so that the tail of the launch is made of the short ones."""
    import ctypes
    by_dev = {}
    for j in jobs:
        by_dev.setdefault(j[0].device, []).append(j)
    for lst in by_dev.values():
        lst = sorted(lst, key=lambda j: -(j[4] // j[7]))
        arr = (_lib.TnJob * len(lst))()
        nbytes = flops = pbytes = 0
        for a, (dy, x, part, dbp, M, N, K, splits, zlo, zhi) in zip(arr, lst):
            a.dy, a.x, a.partial, a.dbias_partial = dy.data_ptr(), x.data_ptr(), part.data_ptr(), (None if dbp is None else dbp.data_ptr())
            a.M, a.N, a.K, a.splits, a.partial_dtype, a.zero_lo, a.zero_hi = M, N, K, splits, dtype_code(part), zlo, zhi
            nbytes += 2 * (M * K + M * N) + 4 * N * K
            flops += 2 * M * K * N
            pbytes += part.element_size() * splits * N * K if splits > 1 else 0
        call("pswin_gemm_tn_ring_jobs", lst[0][0], ctypes.cast(arr, ctypes.c_void_p), len(lst), algo_bytes=nbytes, algo_flops=flops,
             timed_as="pswin_gemm_tn_ring", partial_bytes=pbytes)


def _launch_queue(q):
    jobs, tjobs, wjobs = q["jobs"], q["table_jobs"], q["wgrad_jobs"]
    q["jobs"], q["table_jobs"], q["wgrad_jobs"] = [], [], []
    if wjobs:
        _launch_wgrads(wjobs)                           # the reductions below sum their partial slabs
    if tjobs and DEFER_TABLE_PARTIALS:
        _launch_table_grads(tjobs, 1)                   # partial-row sums of every attention module's dScore tiles, one launch
    if jobs:
        _launch_reductions(jobs)
    if tjobs:
        _launch_table_grads(tjobs, 4)                   # per-bin sums from the partial-row sums the reductions just wrote


def flush_reductions(task=None):
    """Issue the reductions queued by one backward pass (runs by itself when that pass ends); task=None: all of them."""
    keys = list(_ReduceQueue.tasks) if task is None else [task]
    for k in keys:
        q = _ReduceQueue.tasks.pop(k, None)
        if q is not None:
            _launch_queue(q)


def _has_hooks(p):
    return bool(getattr(p, "_post_accumulate_grad_hooks", None)) or bool(getattr(p, "_backward_hooks", None))


def _deferring(owners=()):
    """The job queue of the running backward pass if the reduction that produces the gradients of `owners` (parameters)
    may be postponed to the end of that pass (see set_deferred_reductions; the end-of-pass callback is armed on first
    use), else None = launch now.  Without owners the destination is unknown and nothing is postponed."""
    task = _graph_task_id() if _ReduceQueue.enabled else -1
    owners = [o for o in owners if o is not None]
    if task == -1 or not owners:
        return None
    if any((not o.is_leaf) or o.grad is not None or _has_hooks(o) for o in owners):
        return None                  # a non-leaf "owner" feeds further autograd nodes right away
    q = _ReduceQueue.tasks.get(task)
    if q is None:
        q = _ReduceQueue.tasks[task] = {"jobs": [], "table_jobs": [], "wgrad_jobs": [], "owners": set()}
        _ReduceQueue.trim(_ReduceQueue.tasks)
        torch.autograd.Variable._execution_engine.queue_callback(lambda: flush_reductions(task))
    keys = [o.data_ptr() for o in owners]
    if any(k in q["owners"] for k in keys):
        _launch_queue(q)             # second gradient of a parameter in one pass: autograd adds it to the first one now
        return None
    q["owners"].update(keys)
    return q


def flush_if_pending(owners=()):
    """A gradient of `owners` is about to be returned to autograd WITHOUT going through the queue (a weight gradient small enough
    for one launch): if an earlier use of the same parameter in this pass left a postponed (still unfilled) gradient in the queue,
    autograd would add the two at once -- issue the queue first.  (A module applied to one large and one small input.)"""
    if not _ReduceQueue.enabled:
        return
    q = _ReduceQueue.tasks.get(_graph_task_id())
    if q is not None and any(o is not None and o.data_ptr() in q["owners"] for o in owners):
        _launch_queue(q)


def grad_slot(param):
    """The flat-gradient-buffer view dp.GradReducer(pack=True) reserved for `param` (``param._grad_slot``) if this backward
    pass may write the parameter's gradient straight into it: the parameter holds no gradient yet (nothing to accumulate
    into), has no hooks, and the slot has not been handed out earlier in the same pass (a weight used twice gets a
    private buffer the second time and autograd adds the two).  None otherwise."""
    slot = getattr(param, "_grad_slot", None)
    if slot is None or not param.is_leaf or param.grad is not None or slot.device != param.device or _has_hooks(param):
        return None
    task = _graph_task_id()
    if task == -1:
        return None
    used = _ReduceQueue.slots.get(task)
    if used is None:
        used = _ReduceQueue.slots[task] = set()
        _ReduceQueue.trim(_ReduceQueue.slots)
    if slot.data_ptr() in used:
        return None
    used.add(slot.data_ptr())
    return slot


def sum_rows(src, rows, cols, ld=None, col_offset=0, out=None, owners=()):
    """f32 [cols]: out[c] = sum_{r < rows} src.flatten()[r * ld + col_offset + c] in a fixed order (pswin_reduce_jobs).
    owners: the parameters whose gradient the result is; inside a backward pass with set_deferred_reductions(True) the
    launch is then postponed to the end of that pass when that is safe (_deferring).
    out: an existing contiguous f32 buffer of `cols` elements to write (see grad_slot); a fresh view of it is returned."""
    ld = cols if ld is None else ld
    if not src.is_contiguous():
        raise PswinError("sum_rows expects a contiguous source")
    if out is not None:
        if out.dtype != torch.float32 or out.numel() != cols or not out.is_contiguous():
            raise PswinError("sum_rows: `out` must be a contiguous float32 buffer of `cols` elements")
        job = (src, col_offset * src.element_size(), dtype_code(src), rows, cols, ld, out)
        q = _deferring(owners)
        if q is not None:
            q["jobs"].append(job)
        else:
            _launch_reductions([job])
        return out.view(cols)
    out = torch.empty(cols, dtype=torch.float32, device=src.device)
    job = (src, col_offset * src.element_size(), dtype_code(src), rows, cols, ld, out)
    q = _deferring(owners)
    if q is None:
        _launch_reductions([job])
        return out
    q["jobs"].append(job)
    # the queue keeps `out` alive until the launch; hand autograd a fresh view so that AccumulateGrad can still adopt
    # the buffer as param.grad (it clones tensors that have other owners)
    return out.view(cols)


# ------------------------------------------------------------------------------------------------
# static (input independent) tables, cached per feature-map shape and device
# ------------------------------------------------------------------------------------------------
def window_maps(pano, H, W, shift, device):
    """(map int32 [nW*49], inv int32 [H*W], nW): WindowTransition + pad_x + window_partition as an index map
    (HOT:376-409, 486-491, 64-75) and its inverse (HOT:78-92, 516-528)."""
    key = ("map", bool(pano), H, W, shift, _dev_key(device))
    if key not in _CACHE:
        mode = MODE_PANO if pano else MODE_PLANAR
        _, _, nW = _lib.window_grid(mode, H, W)
        wmap = torch.empty(nW * WTOK, dtype=torch.int32, device=device)
        inv = torch.empty(H * W, dtype=torch.int32, device=device)
        call("pswin_window_map", wmap, mode, H, W, shift, ptr(wmap), ptr(inv))
        _CACHE[key] = (wmap, inv, nW)
    return _CACHE[key]


def window_pads(pano, H, W, shift, device):
    """int32 [nW*49 - H*W]: the window slots no token maps to (the zero rows of pad_x, HOT:486-491), ascending"""
    key = ("pads", bool(pano), H, W, shift, _dev_key(device))
    if key not in _CACHE:
        wmap, _, _ = window_maps(pano, H, W, shift, device)
        _CACHE[key] = torch.nonzero(wmap < 0).flatten().to(torch.int32).contiguous()
    return _CACHE[key]


def identity_map(S, device):
    """int32 arange(S): the row movers with this map are plain (scaled, residual-added, dtype-converting) row copies."""
    key = ("ident", S, _dev_key(device))
    if key not in _CACHE:
        _CACHE[key] = torch.arange(S, dtype=torch.int32, device=device)
    return _CACHE[key]


def planar_mask(H, W, shift, device):
    """BasicLayer._get_attention_mask (HOT:664-688): f32 [nW, 49, 49] of 0 / -100."""
    key = ("mask", H, W, shift, _dev_key(device))
    if key not in _CACHE:
        _, _, nW = _lib.window_grid(MODE_PLANAR, H, W)
        mask = torch.empty(nW, WTOK, WTOK, dtype=torch.float32, device=device)
        call("pswin_planar_mask", mask, H, W, shift, ptr(mask))
        _CACHE[key] = mask
    return _CACHE[key]


def uv_grid(H, W, device):
    """make_uv_hw2 (HOT:153-189): f32 [H*W, 2]."""
    key = ("uv", H, W, _dev_key(device))
    if key not in _CACHE:
        uv = torch.empty(H * W, 2, dtype=torch.float32, device=device)
        call("pswin_uv_grid", uv, H, W, ptr(uv))
        _CACHE[key] = uv
    return _CACHE[key]


def abs_pos_features(H, W, device):
    """xyzuv features of _pano_abs_position (HOT:926-932): f32 [H*W, 5]."""
    key = ("xyzuv", H, W, _dev_key(device))
    if key not in _CACHE:
        uv = uv_grid(H, W, device)
        feat = torch.empty(H * W, 5, dtype=torch.float32, device=device)
        call("pswin_abs_pos_features", feat, ptr(uv), H * W, ptr(feat))
        _CACHE[key] = feat
    return _CACHE[key]


def gather_uv(uv, wmap):
    out = torch.empty(wmap.numel(), 2, dtype=torch.float32, device=uv.device)
    call("pswin_gather_uv", uv, ptr(uv), ptr(wmap), wmap.numel(), ptr(out))
    return out


def haversine_windows(uv1, uv2):
    """haversine22 per window (lzx/models/great_circle.py:71-86): [n, 49, 2] x [n, 49, 2] -> [n, 49, 49]."""
    uv1, uv2 = uv1.contiguous().float(), uv2.contiguous().float()
    n = uv1.numel() // (2 * WTOK)
    dist = torch.empty(n, WTOK, WTOK, dtype=torch.float32, device=uv1.device)
    call("pswin_haversine_windows", dist, ptr(uv1), ptr(uv2), n, ptr(dist))
    return dist


def window_dist(H, W, shift, device):
    """Great-circle distance table of a pano block: [nW, 49, 49], identical for every image and step."""
    key = ("dist", H, W, shift, _dev_key(device))
    if key not in _CACHE:
        wmap, _, nW = window_maps(True, H, W, shift, device)
        uvw = gather_uv(uv_grid(H, W, device), wmap).view(nW, WTOK, 2)
        _CACHE[key] = haversine_windows(uvw, uvw)
    return _CACHE[key]


# ------------------------------------------------------------------------------------------------
# row movers
# ------------------------------------------------------------------------------------------------
def _gather_raw(x, wmap, scale, n_slots, out_dtype):
    B, S, C = x.shape
    win = torch.empty(B, n_slots, C, dtype=out_dtype, device=x.device)
    nreal = min(S, n_slots)
    call("pswin_window_gather", x, ptr(x), dtype_code(x), ptr(wmap), ptr(scale), ptr(win), dtype_code(win), B, S,
         n_slots, C, algo_bytes=B * C * (nreal * x.element_size() + n_slots * win.element_size()))
    return win


def _scatter_raw(win, inv, resid, scale, S, out_dtype, bias=None):
    B, n_slots, C = win.shape
    out = torch.empty(B, S, C, dtype=out_dtype, device=win.device)
    call("pswin_window_scatter_add", win, ptr(win), dtype_code(win), ptr(inv), ptr(resid), ptr(scale), ptr(bias), ptr(out),
         dtype_code(out), B, S, n_slots, C,
         algo_bytes=B * S * C * (win.element_size() + out.element_size() * (1 if resid is None else 2)))
    return out


class _WindowGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, wmap, inv, out_dtype):
        ctx.save_for_backward(inv)
        ctx.S, ctx.in_dtype = x.shape[1], x.dtype
        return _gather_raw(x.contiguous(), wmap, None, wmap.numel(), out_dtype)

    @staticmethod
    def backward(ctx, dwin):
        (inv,) = ctx.saved_tensors
        return _scatter_raw(dwin.contiguous(), inv, None, None, ctx.S, ctx.in_dtype), None, None, None


def window_gather(x, wmap, inv, out_dtype=None):
    """[B, S, C] -> [B, nW*49, C]: shift + pad + window partition in one indexed row copy."""
    return _WindowGather.apply(x, wmap, inv, out_dtype or x.dtype)


class _WindowScatterAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, win, resid, wmap, inv, scale, bias, bias_grad_elsewhere):
        ctx.save_for_backward(wmap, scale)
        ctx.win_dtype = win.dtype
        ctx.bias_grad = bias is not None and not bias_grad_elsewhere
        b = None if bias is None else bias.detach().float().contiguous()
        return _scatter_raw(win.contiguous(), inv, resid.contiguous(), scale, resid.shape[1], resid.dtype, b)

    @staticmethod
    def backward(ctx, dout):
        wmap, scale = ctx.saved_tensors
        dout = dout.contiguous()
        dwin = _gather_raw(dout, wmap, scale, wmap.numel(), ctx.win_dtype)
        dbias = None
        if ctx.bias_grad and ctx.needs_input_grad[5]:       # generic path: one extra pass (the blocks avoid it, see below)
            g = dout.float() if scale is None else dout.float() * scale[:, None, None]
            dbias = colsum(g.reshape(-1, g.shape[-1]))
        return dwin, dout, None, None, None, dbias, None


def window_scatter_add(win, resid, wmap, inv, scale=None, bias=None, bias_grad_elsewhere=False):
    """resid + scale_b * (window_reverse(win) + bias): window reverse + crop + reverse shift + DropPath + residual
    (HOT:483, 516-533) in one indexed row copy.  win [B, nW*49, C], resid [B, S, C].

    bias: the bias of the Linear that produced win (so that the GEMM needs no epilogue).  bias_grad_elsewhere=True:
    this op returns no gradient for it; the caller obtains it from layer_norm_gather(..., res_bias=bias), whose
    backward kernel reads the shortcut gradient anyway."""
    return _WindowScatterAdd.apply(win, resid, wmap, inv, scale, bias, bias_grad_elsewhere)


class _PatchMergeGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, H, W, out_dtype):
        B, S, C = x.shape
        ctx.geom = (B, H, W, C, x.dtype)
        H2, W2 = (H + 1) // 2, (W + 1) // 2
        x = x.contiguous()
        out = torch.empty(B, H2 * W2, 4 * C, dtype=out_dtype, device=x.device)
        call("pswin_patch_merge_gather", x, ptr(x), dtype_code(x), ptr(out), dtype_code(out), B, H, W, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, H, W, C, dt = ctx.geom
        dout = dout.contiguous()
        dx = torch.empty(B, H * W, C, dtype=dt, device=dout.device)
        call("pswin_patch_merge_scatter", dout, ptr(dout), dtype_code(dout), ptr(dx), dtype_code(dx), B, H, W, C)
        return dx, None, None, None


def patch_merge_gather(x, H, W, out_dtype=None):
    """PatchMerging's 2x2 strided gather + concat (HOT:560-573): [B, H*W, C] -> [B, ceil(H/2)*ceil(W/2), 4C]."""
    return _PatchMergeGather.apply(x, H, W, out_dtype or x.dtype)


class _LayerNormGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, wmap, inv, out_dtype, passthrough, res_bias, res_scale, add_rows=None):
        B, S, C = x.shape
        x = x.contiguous()
        n_out = S if wmap is None else wmap.numel()
        y = torch.empty(B, n_out, C, dtype=out_dtype, device=x.device)
        mean = torch.empty(B, S, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        if add_rows is None:
            call("pswin_ln_gather_fwd", x, ptr(x), dtype_code(x), ptr(wmap), ptr(gamma), ptr(beta), float(eps), ptr(y),
                 dtype_code(y), ptr(mean), ptr(rstd), B, S, n_out, C,
                 algo_bytes=B * C * (min(S, n_out) * x.element_size() + n_out * y.element_size()))
        else:
            if wmap is not None or add_rows.shape != (S, C) or add_rows.dtype != torch.float32 or passthrough:
                raise PswinError("layer_norm_gather(add_rows=...) takes f32 [S, C] rows, no window map and no passthrough")
            add_rows = add_rows.contiguous()
            call("pswin_ln_gather_fwd_add", x, ptr(x), dtype_code(x), None, ptr(gamma), ptr(beta), float(eps), ptr(add_rows), ptr(y),
                 dtype_code(y), ptr(mean), ptr(rstd), B, S, n_out, C, timed_as="pswin_ln_gather_fwd",
                 algo_bytes=B * C * (S * x.element_size() + n_out * y.element_size()) + 4 * S * C)
        ctx.has_add = add_rows is not None
        ctx.save_for_backward(x, gamma, mean, rstd, inv, res_scale)
        ctx.n_out = n_out
        ctx.want_res_sum = res_bias is not None
        ctx.owners = (gamma, beta, res_bias)
        if passthrough:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, gamma, mean, rstd, inv, res_scale = ctx.saved_tensors
        B, S, C = x.shape
        dx = torch.empty_like(x)
        dres_sum = None
        if dres is not None:
            if x.dtype != torch.float32:
                raise PswinError("the residual passthrough of layer_norm_gather needs an fp32 residual stream")
            dres = dres.float().contiguous()
        if ctx.want_res_sum and dres is None:
            raise PswinError("layer_norm_gather(res_bias=...) needs the passthrough output to be used as the shortcut")
        if dy is None:                                   # only the shortcut was used downstream
            if ctx.want_res_sum:
                g = dres if res_scale is None else dres * res_scale[:, None, None]
                dres_sum = colsum(g.reshape(-1, C), owners=ctx.owners[2:])
            return dres, torch.zeros_like(gamma), torch.zeros_like(gamma), None, None, None, None, None, dres_sum, None, None
        dy = dy.contiguous()
        lib = _lib.load()
        ws = torch.empty(lib.pswin_ln_workspace(B * S, C), dtype=torch.float32, device=x.device)
        nseg = 3 if ctx.want_res_sum else 2
        # partial rows only ([dgamma | dbeta | dres_sum] per block); summed by the grouped reduction
        call("pswin_ln_gather_bwd", x, ptr(dy), dtype_code(dy), ptr(inv), ptr(x), dtype_code(x), ptr(mean), ptr(rstd),
             ptr(gamma), ptr(dres), ptr(res_scale), ptr(dres) if ctx.want_res_sum else None, ptr(dx), None, None, ptr(ws),
             B, S, ctx.n_out, C, algo_bytes=B * S * C * (dy.element_size() + (2 if dres is None else 3) * x.element_size()))
        sums = sum_rows(ws, lib.pswin_ln_partial_rows(B * S, C), nseg * C, owners=ctx.owners)
        if ctx.want_res_sum:
            dres_sum = sums[2 * C:]
        # the added rows' gradient: a per-element sum over the batch (one pass, no framework two-pass reduction)
        dadd = (dy.float().sum(0) if B > 1 else dy[0].float()) if ctx.has_add else None
        return dx, sums[:C], sums[C:2 * C], None, None, None, None, None, dres_sum, None, dadd


def layer_norm_gather(x, gamma, beta, eps, wmap=None, inv=None, out_dtype=None, passthrough=False, res_bias=None,
                      res_scale=None, add_rows=None):
    """LayerNorm over the last dim of x [B, S, C], written through a window map (norm1 + shift + pad + window
    partition, HOT:503-513) or in place order (wmap=None: norm2 / output norms).  Padding slots are zero rows.

    passthrough=True returns (y, x'): x' is x itself, to be used for the residual shortcut (x' + f(y)); the gradient
    that flows back into x' is then added to dx INSIDE the LayerNorm backward kernel instead of by a separate
    accumulation pass over the residual stream (one per block half in the reference's autograd graph).

    res_bias (with passthrough): the bias b of the branch x' + scale_b * (f(y) + b) that window_scatter_add(...,
    bias=b, bias_grad_elsewhere=True) adds; its gradient, sum_{b,t} res_scale[b] * grad(x')[b][t], is accumulated by
    the same backward kernel (which reads grad(x') anyway) and returned here."""
    if add_rows is not None:        # + f32 [S, C] rows after the affine (the absolute position encoding): pswin_ln_gather_fwd_add
        return _LayerNormGather.apply(x, gamma, beta, eps, wmap, inv, out_dtype or x.dtype, passthrough, res_bias, res_scale, add_rows)
    return _LayerNormGather.apply(x, gamma, beta, eps, wmap, inv, out_dtype or x.dtype, passthrough, res_bias, res_scale)


# the window gather (or bf16 cast) of window_scatter_add's backward from inside the LayerNorm backward kernel (pswin_ln_gather_bwd_ex) and
# the next block's norm1 + partition from inside the residual add (pswin_scatter_add_ln_fwd_map); off: separate kernels
LN_FUSED_MOVES = _on("ln_fused_moves")


class _ScatterAddLayerNorm(torch.autograd.Function):
    """window_scatter_add + LayerNorm in one forward pass (pswin_scatter_add_ln_fwd[_map]); the backward pass is the LayerNorm backward
    kernel (with the shortcut gradient folded in, as layer_norm_gather(passthrough=True)) which also writes the window gather of
    window_scatter_add's backward (pswin_ln_gather_bwd_ex) -- or, with ln_fused_moves off / fp32 windows, followed by that gather."""

    @staticmethod
    def forward(ctx, win, resid, wmap, inv, scale, bias, gamma, beta, eps, out_dtype, res_bias, res_scale, in_pads, out_inv, out_n, out_pads):
        B, S, C = resid.shape
        if resid.dtype != torch.float32:
            raise PswinError("scatter_add_layer_norm needs an fp32 residual stream")
        win, resid = win.contiguous(), resid.contiguous()
        x1 = torch.empty_like(resid)
        n_y = S if out_inv is None else int(out_n)
        y = torch.empty(B, n_y, C, dtype=out_dtype, device=resid.device)
        mean = torch.empty(B, S, dtype=torch.float32, device=resid.device)
        rstd = torch.empty_like(mean)
        b = None if bias is None else bias.detach().float().contiguous()
        abytes = B * C * (win.shape[1] * win.element_size() + 8 * S + n_y * y.element_size())
        if out_inv is None:
            call("pswin_scatter_add_ln_fwd", win, ptr(win), dtype_code(win), ptr(inv), ptr(resid), ptr(scale), ptr(b), ptr(x1),
                 ptr(gamma), ptr(beta), float(eps), ptr(y), dtype_code(y), ptr(mean), ptr(rstd), B, S, win.shape[1], C, algo_bytes=abytes)
        else:
            call("pswin_scatter_add_ln_fwd_map", win, ptr(win), dtype_code(win), ptr(inv), ptr(resid), ptr(scale), ptr(b), ptr(x1),
                 ptr(gamma), ptr(beta), float(eps), ptr(y), dtype_code(y), ptr(mean), ptr(rstd), B, S, win.shape[1], C, ptr(out_inv), n_y,
                 ptr(out_pads), n_y - S, algo_bytes=abytes, timed_as="pswin_scatter_add_ln_fwd")
        ctx.save_for_backward(x1, gamma, mean, rstd, wmap, scale, res_scale, inv, in_pads, out_inv)
        ctx.win_dtype, ctx.want_res_sum, ctx.n_in, ctx.n_y = win.dtype, res_bias is not None, win.shape[1], n_y
        ctx.owners = (gamma, beta, res_bias)
        return y, x1

    @staticmethod
    def backward(ctx, dy, dres):
        x1, gamma, mean, rstd, wmap, scale, res_scale, inv, in_pads, out_inv = ctx.saved_tensors
        B, S, C = x1.shape
        if dy is None:
            raise PswinError("scatter_add_layer_norm: the normalised output must be used")
        if ctx.want_res_sum and dres is None:
            raise PswinError("scatter_add_layer_norm(res_bias=...) needs the second output to be used as the shortcut")
        dy = dy.contiguous()
        dres = None if dres is None else dres.float().contiguous()
        dx1 = torch.empty_like(x1)
        lib = _lib.load()
        ws = torch.empty(lib.pswin_ln_workspace(B * S, C), dtype=torch.float32, device=x1.device)
        nseg = 3 if ctx.want_res_sum else 2
        n_pads = ctx.n_in - S
        fused = (LN_FUSED_MOVES and ctx.win_dtype == torch.bfloat16 and C <= 1024 and (inv is not None or n_pads == 0)
                 and (n_pads == 0 or in_pads is not None))
        abytes = B * S * C * (dy.element_size() + (2 if dres is None else 3) * 4)
        if fused:                                            # dwin = scale_b * dx1 through the window map, from the same kernel
            dwin = torch.empty(B, ctx.n_in, C, dtype=torch.bfloat16, device=x1.device)
            call("pswin_ln_gather_bwd_ex", x1, ptr(dy), dtype_code(dy), ptr(out_inv), ptr(x1), dtype_code(x1), ptr(mean), ptr(rstd),
                 ptr(gamma), ptr(dres), ptr(res_scale), ptr(dres) if ctx.want_res_sum else None, ptr(dx1), None, None, ptr(ws),
                 B, S, ctx.n_y, C, ptr(dwin), ptr(inv), ctx.n_in, ptr(scale), ptr(in_pads), n_pads,
                 algo_bytes=abytes + B * ctx.n_in * C * 2, timed_as="pswin_ln_gather_bwd")
        else:
            call("pswin_ln_gather_bwd", x1, ptr(dy), dtype_code(dy), ptr(out_inv), ptr(x1), dtype_code(x1), ptr(mean), ptr(rstd),
                 ptr(gamma), ptr(dres), ptr(res_scale), ptr(dres) if ctx.want_res_sum else None, ptr(dx1), None, None, ptr(ws),
                 B, S, ctx.n_y, C, algo_bytes=abytes)
        sums = sum_rows(ws, lib.pswin_ln_partial_rows(B * S, C), nseg * C, owners=ctx.owners)
        dres_sum = sums[2 * C:] if ctx.want_res_sum else None
        if not fused:
            dwin = _gather_raw(dx1, wmap, scale, wmap.numel(), ctx.win_dtype)      # window_scatter_add's backward
        return dwin, dx1, None, None, None, None, sums[:C], sums[C:2 * C], None, None, dres_sum, None, None, None, None, None


def scatter_add_layer_norm(win, resid, wmap, inv, scale, bias, gamma, beta, eps, out_dtype=None, res_bias=None,
                           res_scale=None, in_pads=None, out=None):
    """(y, x1) with x1 = resid + scale_b * (window_reverse(win) + bias) and y = LayerNorm(x1): window_scatter_add(...,
    bias_grad_elsewhere=True) followed by layer_norm_gather(x1, passthrough=True, res_bias=..., res_scale=...) as ONE
    forward kernel (the shortcut sum is not re-read by a LayerNorm kernel).  x1 is the tensor to use for the next
    shortcut.  The gradient of `bias` is not returned here (it comes from the LayerNorm that produced `resid` with
    res_bias=bias); res_bias / res_scale as in layer_norm_gather.
    inv=None: `win` is in token order (wmap: the identity map).  in_pads: window_pads of the map `win` is laid out by (lets the backward
    kernel write the gathered gradient itself).  out=(out_inv, n_out, out_pads): y is written through the token -> slot map out_inv into
    [B, n_out, C] with zero rows at out_pads -- the NEXT block's norm1 + shift + pad + window partition (layer_norm_gather(x1, ..., wmap,
    inv)) fused into this pass."""
    out_inv, out_n, out_pads = out if out is not None else (None, 0, None)
    return _ScatterAddLayerNorm.apply(win, resid, wmap, inv, scale, bias, gamma, beta, eps, out_dtype or resid.dtype,
                                      res_bias, res_scale, in_pads, out_inv, out_n, out_pads)


class _LayerNormNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, H, W, passthrough):
        B, S, C = x.shape
        x = x.contiguous()
        y = torch.empty(B, C, H, W, dtype=torch.float32, device=x.device)
        mean = torch.empty(B, S, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        call("pswin_ln_nchw_fwd", x, ptr(x), ptr(gamma), ptr(beta), float(eps), ptr(y), ptr(mean), ptr(rstd), B, S, C,
             algo_bytes=2 * x.numel() * 4)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.owners = (gamma, beta)
        if passthrough:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, gamma, mean, rstd = ctx.saved_tensors
        B, S, C = x.shape
        if dy is None:
            return dres, torch.zeros_like(gamma), torch.zeros_like(gamma), None, None, None, None
        dy = dy.float().contiguous()
        if dres is not None:
            dres = dres.float().contiguous()
        dx = torch.empty_like(x)
        lib = _lib.load()
        ws = torch.empty(lib.pswin_ln_workspace(B * S, C), dtype=torch.float32, device=x.device)
        call("pswin_ln_nchw_bwd", x, ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(dres), ptr(dx), None, None,
             ptr(ws), B, S, C, algo_bytes=(3 if dres is None else 4) * x.numel() * 4)
        sums = sum_rows(ws, lib.pswin_ln_partial_rows(B * S, C), 2 * C, owners=ctx.owners)
        return dx, sums[:C], sums[C:], None, None, None, None


class _ScatterAddLayerNormNCHW(torch.autograd.Function):
    """The end of a stage in one autograd node and one forward kernel (pswin_scatter_add_ln_nchw_fwd): x = resid + scale_b * (y + bias)
    (the closing MLP branch of the last block, HOT:536) and the output norm of x written as NCHW (HOT:975-977).
    Returns (normed NCHW map, x).  Backward: ONE pswin_ln_nchw_bwd_ex launch yields d(resid) (with the gradient that reaches x from the
    next stage folded in, as layer_norm_nchw(passthrough=True)) AND d(y) = bf16(scale_b * dx) -- the cast that window_scatter_add's
    backward otherwise runs as a pass of its own.  The bias gradient is obtained elsewhere (norm2's backward, res_bias)."""

    @staticmethod
    def forward(ctx, y, resid, scale, bias, gamma, beta, eps, H, W):
        B, S, C = resid.shape
        b = None if bias is None else bias.detach().float().contiguous()
        y, resid = y.contiguous(), resid.contiguous()
        x = torch.empty_like(resid)
        out = torch.empty(B, C, H, W, dtype=torch.float32, device=x.device)
        mean = torch.empty(B, S, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        call("pswin_scatter_add_ln_nchw_fwd", x, ptr(y), ptr(resid), ptr(scale), ptr(b), ptr(x), ptr(gamma), ptr(beta), float(eps),
             ptr(out), ptr(mean), ptr(rstd), B, S, C, timed_as="pswin_ln_nchw_fwd", algo_bytes=x.numel() * (2 + 3 * 4))
        ctx.save_for_backward(x, gamma, mean, rstd, scale)
        ctx.owners = (gamma, beta)
        ctx.y_dtype = y.dtype
        return out, x

    @staticmethod
    def backward(ctx, dout, dres):
        x, gamma, mean, rstd, scale = ctx.saved_tensors
        B, S, C = x.shape
        if dout is None:                                  # only the residual stream was used downstream
            dx = dres.float().contiguous()
            dy = _gather_raw(dx, identity_map(S, x.device), scale, S, ctx.y_dtype)
            return dy, dx, None, None, torch.zeros_like(gamma), torch.zeros_like(gamma), None, None, None
        dout = dout.float().contiguous()
        if dres is not None:
            dres = dres.float().contiguous()
        dx = torch.empty_like(x)
        ex = torch.empty(B, S, C, dtype=torch.bfloat16, device=x.device)
        lib = _lib.load()
        ws = torch.empty(lib.pswin_ln_workspace(B * S, C), dtype=torch.float32, device=x.device)
        call("pswin_ln_nchw_bwd_ex", x, ptr(dout), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(dres), ptr(dx), None, None,
             ptr(ws), ptr(ex), ptr(scale), B, S, C, timed_as="pswin_ln_nchw_bwd",
             algo_bytes=(3 if dres is None else 4) * x.numel() * 4 + 2 * x.numel())
        sums = sum_rows(ws, lib.pswin_ln_partial_rows(B * S, C), 2 * C, owners=ctx.owners)
        return ex.to(ctx.y_dtype), dx, None, None, sums[:C], sums[C:], None, None, None


def scatter_add_layer_norm_nchw_supported(y, resid):
    """bf16 branch rows, fp32 residual stream, a shape the NCHW LayerNorm kernels tile"""
    return (LN_FUSED_MOVES and y.dtype == torch.bfloat16 and resid.dtype == torch.float32 and y.shape == resid.shape
            and bool(_lib.load().pswin_ln_nchw_supported(resid.shape[1], resid.shape[2])))


def scatter_add_layer_norm_nchw(y, resid, scale, bias, gamma, beta, eps, H, W):
    """(LayerNorm_NCHW(x), x) with x = resid + scale_b * (y + bias): see _ScatterAddLayerNormNCHW"""
    return _ScatterAddLayerNormNCHW.apply(y, resid, scale, bias, gamma, beta, eps, H, W)


def layer_norm_nchw(x, gamma, beta, eps, H, W, passthrough=False):
    """LayerNorm over the channels of x [B, H*W, C] (fp32) returned as a contiguous NCHW map [B, C, H, W]: the output
    norms of the backbone (HOT:975-977) without the separate transpose pass, forward and backward.  Falls back to
    layer_norm_gather + permute when the shape does not fit the kernel's tiling."""
    B, S, C = x.shape
    if x.dtype != torch.float32 or not _lib.load().pswin_ln_nchw_supported(S, C):
        out = layer_norm_gather(x, gamma, beta, eps, passthrough=passthrough)
        y, x2 = out if passthrough else (out, None)
        y = y.float().view(B, H, W, C).permute(0, 3, 1, 2).contiguous()
        return (y, x2) if passthrough else y
    return _LayerNormNCHW.apply(x, gamma, beta, eps, H, W, passthrough)


class _LayerNormPatchMerge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, H, W, out_dtype):
        B, S, C = x.shape
        x = x.contiguous()
        H2, W2 = (H + 1) // 2, (W + 1) // 2
        y = torch.empty(B, H2 * W2, 4 * C, dtype=out_dtype, device=x.device)
        mean = torch.empty(B, H2 * W2, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        call("pswin_ln_patch_merge_fwd", x, ptr(x), dtype_code(x), ptr(gamma), ptr(beta), float(eps), ptr(y),
             dtype_code(y), ptr(mean), ptr(rstd), B, H, W, C)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.owners = (gamma, beta)
        ctx.geom = (H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        B, S, C = x.shape
        H, W = ctx.geom
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        rows = B * ((H + 1) // 2) * ((W + 1) // 2)
        lib = _lib.load()
        ws = torch.empty(lib.pswin_ln_workspace(rows, 4 * C), dtype=torch.float32, device=x.device)
        call("pswin_ln_patch_merge_bwd", x, ptr(dy), dtype_code(dy), ptr(x), dtype_code(x), ptr(mean), ptr(rstd),
             ptr(gamma), ptr(dx), None, None, ptr(ws), B, H, W, C)
        sums = sum_rows(ws, lib.pswin_ln_partial_rows(rows, 4 * C), 8 * C, owners=ctx.owners)
        return dx, sums[:4 * C], sums[4 * C:], None, None, None, None


def layer_norm_patch_merge(x, gamma, beta, eps, H, W, out_dtype=None):
    """PatchMerging's pad + 2x2 strided gather + concat + LayerNorm(4C) (HOT:563-574) in one kernel."""
    return _LayerNormPatchMerge.apply(x, gamma, beta, eps, H, W, out_dtype or x.dtype)


class _BatchNormReLU(torch.autograd.Function):
    """nn.BatchNorm2d (batch or running statistics) + ReLU on a channels-last NCHW tensor (HOT:742-748)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, eps, momentum, train, pre_bias):
        N, C, H, W = y.shape
        rows = y.permute(0, 2, 3, 1)
        assert rows.is_contiguous(), "BatchNorm+ReLU kernel expects a channels-last activation"
        M = N * H * W
        z = torch.empty_like(y)                                     # preserves the channels-last strides
        mean = torch.empty(C, dtype=torch.float32, device=y.device)
        rstd = torch.empty_like(mean)
        ws = torch.empty(_lib.load().pswin_bn_workspace(C), dtype=torch.float32, device=y.device)
        call("pswin_bn_relu_fwd", y, ptr(y), dtype_code(y), ptr(gamma), ptr(beta), ptr(pre_bias), float(eps), float(momentum),
             int(train), ptr(running_mean), ptr(running_var), ptr(z), ptr(mean), ptr(rstd), ptr(ws), M, C)
        ctx.save_for_backward(y, gamma, beta, mean, rstd)
        ctx.train = bool(train)
        ctx.pre_bias_shape = None if pre_bias is None else pre_bias.shape
        return z

    @staticmethod
    def backward(ctx, dz):
        y, gamma, beta, mean, rstd = ctx.saved_tensors
        N, C, H, W = y.shape
        dz = dz.contiguous(memory_format=torch.channels_last)
        dy = torch.empty_like(y)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        ws = torch.empty(_lib.load().pswin_bn_workspace(C), dtype=torch.float32, device=y.device)
        call("pswin_bn_relu_bwd", y, ptr(dz), ptr(y), dtype_code(y), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd),
             int(ctx.train), ptr(dy), ptr(dgamma), ptr(dbeta), ptr(ws), N * H * W, C)
        # a constant added in front of batch-statistics BN has an exactly zero gradient; in eval mode it is sum(dy)
        dpre = None
        if ctx.pre_bias_shape is not None:
            dpre = torch.zeros(ctx.pre_bias_shape, dtype=torch.float32, device=y.device) if ctx.train else \
                colsum(dy.permute(0, 2, 3, 1).reshape(-1, C))
        return dy, dgamma, dbeta, None, None, None, None, None, dpre


def batch_norm_relu(y, bn, training, pre_bias=None):
    """ReLU(BatchNorm2d(y + pre_bias)) for a channels-last y with the parameters / buffers of the nn.BatchNorm2d
    module `bn`; pre_bias (the convolution bias) is never materialised on the activation."""
    use_batch = training or bn.running_mean is None
    if use_batch and bn.num_batches_tracked is not None and training:
        bn.num_batches_tracked.add_(1)
    momentum = 0.1 if bn.momentum is None else bn.momentum
    return _BatchNormReLU.apply(y, bn.weight, bn.bias, bn.running_mean if (training or not use_batch) else None,
                                bn.running_var if (training or not use_batch) else None, bn.eps, momentum, use_batch,
                                pre_bias)


def colsum(x2d, zero_cols=None, owners=()):
    """fp32 column sums of a [M, N] matrix (N % 8 == 0): bias gradients and split-K partial reductions.
    zero_cols=(lo, hi): columns known to sum to zero (not read on the large-matrix path; see pswin_colsum_skip).
    owners: see sum_rows."""
    x2d = x2d.contiguous()
    M, N = x2d.shape
    if M <= 4096:
        return sum_rows(x2d, M, N, owners=owners)
    n_ws = _lib.load().pswin_colsum_workspace(M, N, dtype_code(x2d))
    ws = torch.empty(n_ws, dtype=torch.float32, device=x2d.device)
    if zero_cols is None:
        call("pswin_colsum", x2d, ptr(x2d), dtype_code(x2d), M, N, None, ptr(ws))      # first stage: partial rows
    else:
        call("pswin_colsum_skip", x2d, ptr(x2d), dtype_code(x2d), M, N, int(zero_cols[0]), int(zero_cols[1]), ptr(ws))
    return sum_rows(ws, n_ws // N, N, owners=owners)


def colsum_channels(dy):
    """f32 [C]: the sum of an NCHW gradient over batch and pixels (a convolution's bias gradient) as a fixed-order column sum of its
    channels-last rows, for ANY channel count: rows narrower than / not a multiple of 8 columns (the RPN's 3- and 12-channel heads)
    are zero-padded to the next multiple of 8 first, so that no framework two-pass reduction -- the kind that returns stale results
    from the second replay of a captured hipGraph -- is left on the path."""
    C = dy.shape[1]
    if dy.dtype not in (torch.bfloat16, torch.float32):
        dy = dy.float()
    rows = dy.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(-1, C)
    if C % 8:
        rows = F.pad(rows, (0, 8 - C % 8))
    return colsum(rows)[:C]


def gemm_nt_supported(x2d, n_out):
    """bf16 rows x [n_out, K] weight on the tiled HIP GEMM (pswin_gemm_nt): the Linear layers of stages 1-3"""
    return (x2d.dtype == torch.bfloat16 and x2d.is_cuda and x2d.dim() == 2
            and bool(_lib.load().pswin_gemm_nt_supported(x2d.shape[0], x2d.shape[1], n_out)))


def gemm_nt_tile(M, K, N):
    """Row-tile height (64 / 128) with which pswin_gemm_nt computes [M, K] x [N, K]^T, or 0 = leave it to the library.
    From profiles/r02_gemm_nt_vs_library.txt (MI355X, PanoSwin-T shapes at batch 8) and the in-step A/Bs of round 4
    (profiles/r04_ab_runs.json: gemm_nt_*): the HIP kernel wins wherever the rows fill the chip (M >= 8192: stages 1-2), on the narrow
    outputs with a short contraction at 4-6 k rows, and wherever 128-row tiles still give every CU two rounds of work (stage 3's qkv / fc1
    at batch 8); 128-row tiles then, or when they make the launch fit the chip ONCE (<= 256 tiles: the kernel's four-stage form,
    csrc/pswin_gemm_nt.hip) while 64-row tiles would not; 64-row tiles otherwise."""
    if not GEMM_NT or not bool(_lib.load().pswin_gemm_nt_supported(M, K, N)):
        return 0
    t128 = -(-M // 128) * (N // 192)
    if M < 8192 and not (N <= 768 and K <= 1536) and t128 < 512:
        return 0
    rows = gemm_nt_rows(M, N)
    if rows == 64:
        # 64-row tiles that spill into a second, mostly empty round of the 512 tile slots (two workgroups per CU): 96-row tiles in one
        t64, t96 = -(-M // 64) * (N // 192), -(-M // 96) * (N // 192)
        if 512 < t64 <= 768 and t96 <= 512:
            return 96
    return rows


def gemm_nt_rows(M, N):
    """the row-tile height for a product that runs on pswin_gemm_nt in any case (the fused GELU forms have no library counterpart)"""
    t64, t128 = -(-M // 64) * (N // 192), -(-M // 128) * (N // 192)
    if t128 < 512 and t64 > 256 and t128 <= 256:
        return 128
    return 128 if t128 >= 512 else 64


def transpose_weights(pairs):
    """dst = src^T for every (src [R, C], dst [C, R]) pair of contiguous bf16 matrices, one launch (pswin_transpose_jobs)."""
    import ctypes
    if not pairs:
        return
    arr = (_lib.TransposeJob * len(pairs))()
    for a, (src, dst) in zip(arr, pairs):
        assert src.dtype == dst.dtype == torch.bfloat16 and src.is_contiguous() and dst.is_contiguous()
        assert dst.shape == (src.shape[1], src.shape[0])
        a.src, a.dst, a.rows, a.cols = src.data_ptr(), dst.data_ptr(), src.shape[0], src.shape[1]
    call("pswin_transpose_jobs", pairs[0][0], ctypes.cast(arr, ctypes.c_void_p), len(pairs))


# the three-stage ring kernel for weight gradients (pswin_gemm_tn_ring, round 3); off: library batched GEMMs
GEMM_TN_RING = _on("gemm_tn_ring")
# partial slabs of the row splits in bf16 (what the library's batched GEMM writes; half the slab traffic), f32 for a single split
GEMM_TN_RING_BF16 = True
# the Linear's bias gradient (column sums of dy) from the weight-gradient launch instead of a pass of its own; off: pswin_colsum
GEMM_TN_RING_BIAS = _on("gemm_tn_ring_bias")


# Weight gradients held back to the end of the backward pass and issued as one launch per tile geometry (pswin_gemm_tn_ring_jobs) when
# the parameter-gradient reductions are deferred too (set_deferred_reductions): a workgroup then contracts GROUPED_WGRAD_ROWS rows
# instead of M / (256 / tiles).  off: one launch per weight gradient, where autograd produces it
GROUPED_WGRAD = _on("grouped_wgrad")
GROUPED_WGRAD_ROWS = 2048                 # swept 1,024 / 2,048 / 4,096: profiles/r04_ab_runs.json


def grouped_wgrad_splits(M):
    """row splits of one weight gradient inside the grouped launch: about GROUPED_WGRAD_ROWS rows per workgroup, in counts that divide
    over the 8 XCDs (1, 2, 4 or a multiple of 8: the kernel gives a split to 8, 4, 2 XCDs or one, csrc/pswin_gemm_tn.hip)"""
    s = max(1, min(M // 64, (M + GROUPED_WGRAD_ROWS // 2) // GROUPED_WGRAD_ROWS))
    if s >= 6:
        s = max(8, (s + 4) // 8 * 8)
    elif s >= 3:
        s = 4
    return max(1, min(M // 64, s))


def queue_weight_gradient(dy, x, weight, bias, zero_cols):
    """dW = dy^T x (and the bias gradient, the column sums of dy) through the grouped end-of-pass launch, if this backward pass may
    postpone them (see set_deferred_reductions): returns (dw f32 [N, K], db f32 [N] or None) -- tensors that are FILLED when the pass
    ends -- or None = launch now.  The operands stay referenced by the queue until then."""
    M, N = dy.shape
    K = x.shape[1]
    if not (GROUPED_WGRAD and _ReduceQueue.enabled and dy.dtype == torch.bfloat16 and gemm_tn_ring_splits(M, N, K) > 0):
        return None
    q = _deferring((weight, bias))
    if q is None:
        return None
    splits = grouped_wgrad_splits(M)
    x = x.contiguous()
    slot = grad_slot(weight)
    dev = x.device
    out = slot if slot is not None else torch.empty(N * K, dtype=torch.float32, device=dev)
    if splits == 1:
        part = out                                        # the finished f32 gradient straight from the kernel: nothing to reduce
    else:
        part = torch.empty(splits, N, K, dtype=torch.bfloat16 if GEMM_TN_RING_BF16 else torch.float32, device=dev)
        q["jobs"].append((part, 0, dtype_code(part), splits, N * K, N * K, out))
    dbp = db = None
    if bias is not None:
        db = torch.empty(N, dtype=torch.float32, device=dev)
        dbp = db if splits == 1 else torch.empty(splits, N, dtype=torch.float32, device=dev)
        if splits > 1:
            q["jobs"].append((dbp, 0, F32, splits, N, N, db))
        db = db.view(N)                                   # fresh views: see sum_rows
    zlo, zhi = (0, 0) if zero_cols is None else (int(zero_cols[0]), int(zero_cols[1]))
    q["wgrad_jobs"].append((dy, x, part, dbp, M, N, K, splits, zlo, zhi))
    return out.view(N, K), db


def gemm_tn_ring_splits(M, N, K):
    """Row splits for pswin_gemm_tn_ring on dy [M, N], x [M, K], or 0 = not for this shape.  ONE rule per configuration, whether the
    product is launched alone or inside the grouped launch (the two modes then run the same arithmetic: the deferred step stays bit-equal
    to the immediate one): about GROUPED_WGRAD_ROWS rows per workgroup by default; with grouped_wgrad disabled the round-3 rule, one
    workgroup per CU and launch."""
    if not GEMM_TN_RING or M < 512 or not bool(_lib.load().pswin_gemm_tn_ring_supported(M, N, K)):
        return 0
    if GROUPED_WGRAD:
        return grouped_wgrad_splits(M)
    return int(_lib.load().pswin_gemm_tn_ring_splits(M, N, K, 0))


def gemm_tn_ring(dy, x, splits, out_dtype=torch.float32, bias_sums=False, zero_cols=None):
    """[splits, N, K] partial sums of dy^T x over `splits` row ranges (dy [M, N], x [M, K] bf16) in f32 or bf16.
    Roofline bookkeeping: algorithmic bytes = dY and X read once + the f32 gradient written once, 2 (MK + MN) + 4 NK; the
    split partial slabs are an implementation artefact and are reported separately (partial_bytes), never as algorithmic traffic.
    bias_sums=True: also the f32 [splits, N] column sums of dy per row range (the Linear's bias-gradient partials, from the same
    launch; zero_cols=(lo, hi) columns written as zeros) -> (partials, bias partials)."""
    dy, x = dy.contiguous(), x.contiguous()
    M, N = dy.shape
    K = x.shape[1]
    part = torch.empty(splits, N, K, dtype=out_dtype, device=x.device)
    if not bias_sums:
        call("pswin_gemm_tn_ring", x, ptr(dy), ptr(x), ptr(part), dtype_code(part), M, N, K, int(splits),
             algo_bytes=2 * (M * K + M * N) + 4 * N * K, algo_flops=2 * M * K * N, partial_bytes=part.element_size() * splits * N * K)
        return part
    dbp = torch.empty(splits, N, dtype=torch.float32, device=x.device)
    zlo, zhi = (0, 0) if zero_cols is None else (int(zero_cols[0]), int(zero_cols[1]))
    call("pswin_gemm_tn_ring_bias", x, ptr(dy), ptr(x), ptr(part), dtype_code(part), ptr(dbp), zlo, zhi, M, N, K, int(splits),
         algo_bytes=2 * (M * K + M * N) + 4 * N * K, algo_flops=2 * M * K * N, timed_as="pswin_gemm_tn_ring",
         partial_bytes=part.element_size() * splits * N * K)
    return part, dbp


def gemm_nt(x2d, w, bias=None, tile_m=0, out_f32=False):
    """y = x2d @ w^T (+ bias): x2d [M, K] bf16, w [N, K] bf16, bias f32 [N] or None -> [M, N] bf16 (out_f32: f32, pswin_gemm_nt_f32)."""
    x2d, w = x2d.contiguous(), w.contiguous()
    M, K = x2d.shape
    N = w.shape[0]
    b = None if bias is None else bias.detach().float().contiguous()
    if out_f32:
        y = torch.empty(M, N, dtype=torch.float32, device=x2d.device)
        call("pswin_gemm_nt_f32", x2d, ptr(x2d), ptr(w), ptr(b), ptr(y), M, K, N, int(tile_m), timed_as="pswin_gemm_nt",
             algo_bytes=2 * (M * K + N * K) + 4 * M * N, algo_flops=2 * M * K * N)
        return y
    y = torch.empty(M, N, dtype=torch.bfloat16, device=x2d.device)
    call("pswin_gemm_nt", x2d, ptr(x2d), ptr(w), ptr(b), ptr(y), M, K, N, int(tile_m),
         algo_bytes=2 * (M * K + M * N + N * K), algo_flops=2 * M * K * N)
    return y


def rows_addressable(M, width):
    """The streaming kernels address their row operands through 32-bit buffer offsets: M rows of `width` bf16 elements must stay below
    the 0xFFFFFF00 sentinel (the launchers return PSWIN_ERR_ARG otherwise -- the *_supported predicates below check it first, so an
    oversize batch falls through to the next path instead of raising)."""
    return M * width * 2 < 0xFFFFFF00


def skinny_gemm_shape_ok(M, K, N):
    return M >= 4096 and rows_addressable(M, max(K, N)) and bool(_lib.load().pswin_gemm_skinny_supported(K, N))


def skinny_gemm_supported(x2d, n_out):
    """bf16 rows x a small weight: the shapes pswin_gemm_skinny is instantiated for (stage-0 projections, stage-1 proj)"""
    return x2d.dtype == torch.bfloat16 and x2d.is_cuda and skinny_gemm_shape_ok(x2d.shape[0], x2d.shape[1], n_out)


def skinny_gemm(x2d, w, bias=None, transpose_w=False):
    """y = x2d @ W^T (+ bias) through the streaming HIP GEMM (weights resident in LDS).  transpose_w=False: w is the
    nn.Linear weight [N, K]; transpose_w=True: w is [K, N] and y = x2d @ w (the data gradient with the same weight)."""
    x2d = x2d.contiguous()
    w = w.contiguous()
    M, K = x2d.shape
    N = w.shape[1] if transpose_w else w.shape[0]
    y = torch.empty(M, N, dtype=torch.bfloat16, device=x2d.device)
    b = None if bias is None else bias.detach().float().contiguous()
    call("pswin_gemm_skinny", x2d, ptr(x2d), ptr(w), ptr(b), ptr(y), M, K, N, int(transpose_w),
         algo_bytes=2 * M * (K + N))
    return y


def _pick_split(M, n_tiles):
    """Number of K-splits for a weight-gradient GEMM with contraction length M and n_tiles output tiles: a divisor
    of M that gives hipBLASLt about a thousand independent tiles while each split keeps >= 512 rows."""
    want = max(1, min(M // 512, -(-1024 // n_tiles)))
    best = 1
    for c in range(1, min(M, 4 * want) + 1):
        if M % c == 0 and abs(c - want) < abs(best - want):
            best = c
    return best


class _LinearSplitK(torch.autograd.Function):
    """y = x W^T + b with bf16 operands and fp32 master parameters.

    Forward and dX are plain hipBLASLt GEMMs.  dW = dY^T X contracts over up to 275k window tokens into an output of
    a few dozen tiles, which a single GEMM launch maps onto a few dozen workgroups (measured 470-535 us for the
    stage-0 shapes on MI355X against a 20-50 us HBM floor); it is issued as a batched GEMM over row chunks (an
    explicit split-K, 55-75 us) whose fp32 partial sum also removes the bf16 -> fp32 gradient cast."""

    @staticmethod
    def forward(ctx, x, weight, bias, w_lp, b_lp, zero_bias_cols=None, w_lp_t=None, out_f32=False):
        # out_f32: an f32 result where the tiled HIP GEMM runs (its epilogue writes it), a cast of the bf16 result elsewhere
        # w_lp / b_lp: this step's bf16 copies of the fp32 master parameters (refreshed by ONE multi-tensor cast per
        # forward, see SimplePanoSwinTransformer._refresh_lowp); gradients go to the fp32 masters.
        wb = w_lp if w_lp is not None else weight.to(x.dtype)
        ctx.save_for_backward(x, wb, w_lp_t)            # w_lp_t: this step's [K, N] copy of wb (data gradient on pswin_gemm_nt) or None
        ctx.has_bias = bias is not None
        ctx.zero_bias_cols = zero_bias_cols
        ctx.weight, ctx.bias = weight, bias                 # for grad_slot / owners: dW may be summed straight into its flat slot
        if skinny_gemm_supported(x, wb.shape[0]):       # stage-0 shapes: streaming HIP GEMM, weight resident in LDS
            y = skinny_gemm(x, wb, bias)
            return y.float() if out_f32 else y
        tile = gemm_nt_tile(x.shape[0], x.shape[1], wb.shape[0]) if x.dtype == torch.bfloat16 else 0
        if tile:                                        # stages 1-3: tiled HIP GEMM where it beats the library kernel
            return gemm_nt(x, wb, bias, tile, out_f32=out_f32)
        bb = None if bias is None else (b_lp if b_lp is not None else bias.to(x.dtype))
        M, K, N = x.shape[0], x.shape[1], wb.shape[0]
        with _lib.timed("lib_gemm_fwd", 2 * (M * K + M * N + N * K), 2 * M * K * N):
            y = F.linear(x, wb, bb)
        return y.float() if out_f32 else y

    @staticmethod
    def backward(ctx, dy):
        x, wb, wbt = ctx.saved_tensors
        if dy.dtype != x.dtype:                             # out_f32: the gradient arrives in fp32
            dy = dy.to(x.dtype)
        dx, dw, db = linear_backward(x, wb, dy, ctx.weight, ctx.bias if ctx.has_bias else None, ctx.zero_bias_cols,
                                     ctx.needs_input_grad[0], wbt)
        return dx, dw, db, None, None, None, None, None


def linear_backward(x, wb, dy, weight, bias, zero_bias_cols, need_dx, wbt=None):
    """Gradients of y = x wb^T (+ bias) for bf16 rows x [M, K], wb [N, K]: (dx or None, dW f32 [N, K], dbias f32 [N] or
    None).  weight / bias: the fp32 master parameters the gradients belong to (grad_slot / deferred reductions).
    wbt: wb^T [K, N] if the caller keeps one (dx then runs on pswin_gemm_nt where that is the faster kernel)."""
    dy = dy.contiguous()
    M, N = dy.shape
    K = x.shape[1]
    dx = None
    if need_dx:
        # data gradient with the streaming kernel (weight transposed while it is staged) where that beats the library:
        # the three stage-0 shapes with 96 output columns and the stage-0 fc2 (96 -> 384 columns)
        tile = gemm_nt_tile(M, N, K) if (wbt is not None and dy.dtype == torch.bfloat16) else 0
        if K == 96 and N in (96, 288, 384) and skinny_gemm_supported(dy, K):
            dx = skinny_gemm(dy, wb, None, transpose_w=True)
        elif tile:
            dx = gemm_nt(dy, wbt, None, tile)
        else:
            with _lib.timed("lib_gemm_dgrad", 2 * (M * K + M * N + N * K), 2 * M * K * N):
                dx = dy @ wb
    queued = queue_weight_gradient(dy, x, weight, bias, zero_bias_cols)
    if queued is not None:
        return dx, queued[0], queued[1]
    sp = 0
    rs = gemm_tn_ring_splits(M, N, K) if dy.dtype == torch.bfloat16 else 0
    db_part = None
    if rs:                                                   # the ring-pipelined HIP weight-gradient kernel; the bias gradient rides along
        pdt = torch.bfloat16 if (GEMM_TN_RING_BF16 and rs > 1) else torch.float32
        if bias is not None and GEMM_TN_RING_BIAS:
            part, db_part = gemm_tn_ring(dy, x, rs, pdt, bias_sums=True, zero_cols=zero_bias_cols)
        else:
            part = gemm_tn_ring(dy, x, rs, pdt)
        ch, sp = rs, rs
    else:
        ch = _pick_split(M, -(-N // 64) * -(-K // 64))
        with _lib.timed("lib_gemm_wgrad", 2 * (M * K + M * N) + 4 * N * K, 2 * M * K * N):
            if ch > 1:
                part = torch.bmm(dy.view(ch, M // ch, N).transpose(1, 2), x.view(ch, M // ch, K))
            else:
                dw = (dy.t() @ x).float()
    if ch > 1:
        dw = sum_rows(part, ch, N * K, out=grad_slot(weight), owners=(weight,)).view(N, K)
    else:
        if sp:
            dw = part.view(N, K).float()
        flush_if_pending((weight,))                          # an immediate gradient behind a postponed one of the same weight
    if db_part is not None:
        db = sum_rows(db_part, db_part.shape[0], N, owners=(bias,))
    else:
        db = colsum(dy, zero_bias_cols, owners=(bias,)) if bias is not None else None
    return dx, dw, db


def linear(x, lin, cd, use_bias=True, zero_bias_cols=None, out_f32=False):
    """nn.Linear on rows.  fp32: F.linear (parity path).  bf16: split-K weight gradient, fp32 parameter gradients.
    use_bias=False: the caller applies lin.bias itself (fused into the next row kernel).  zero_bias_cols=(lo, hi): output
    columns whose gradient sums to zero over the rows (their bias gradient is zero; the column sum skips them)."""
    if cd == torch.float32:
        return F.linear(x.float(), lin.weight, lin.bias if use_bias else None)
    shp = x.shape
    x2 = x.to(cd).reshape(-1, shp[-1])
    lp = lin.__dict__.get("_lowp")
    w_lp, b_lp = lp if lp is not None else (None, None)
    bias = lin.bias if use_bias else None
    return _LinearSplitK.apply(x2, lin.weight, bias, w_lp, b_lp if use_bias else None, zero_bias_cols,
                               lin.__dict__.get("_lowp_t"), out_f32).view(*shp[:-1], lin.weight.shape[0])


class _Fc1Gelu(torch.autograd.Function):
    """h = gelu(x W1^T + b1) for the stage-0 Mlp in ONE streaming pass (pswin_fc1_gelu_fwd); the backward pass recomputes
    the pre-activation (K = 96) instead of storing it: dy = dh gelu'(.), db1 = column sums of dy, dx = dy W1 (streaming
    GEMM), dW1 = dy^T x (library batched GEMM over row chunks + fixed-order column sum)."""

    @staticmethod
    def forward(ctx, x, weight, bias, w_lp):
        wb = w_lp if w_lp is not None else weight.to(x.dtype)
        x = x.contiguous()
        M, K = x.shape
        N = wb.shape[0]
        h = torch.empty(M, N, dtype=x.dtype, device=x.device)
        b = bias.detach().float().contiguous()
        call("pswin_fc1_gelu_fwd", x, ptr(x), ptr(wb), ptr(b), ptr(h), M, K, N, algo_bytes=2 * M * (K + N))
        ctx.save_for_backward(x, wb, b)
        ctx.weight, ctx.bias = weight, bias
        return h

    @staticmethod
    def backward(ctx, dh):
        x, wb, b = ctx.saved_tensors
        M, K = x.shape
        N = wb.shape[0]
        dh = dh.to(x.dtype).contiguous()
        dy = torch.empty_like(dh)
        lib = _lib.load()
        ws = torch.empty(lib.pswin_fc1_gelu_workspace(N), dtype=torch.float32, device=x.device)
        call("pswin_fc1_gelu_bwd", x, ptr(x), ptr(wb), ptr(b), ptr(dh), ptr(dy), None, ptr(ws), M, K, N,
             algo_bytes=2 * M * (K + 2 * N))
        db = sum_rows(ws, lib.pswin_fc1_gelu_partial_rows(M), N, owners=(ctx.bias,))
        dx = skinny_gemm(dy, wb, None, transpose_w=True) if ctx.needs_input_grad[0] else None
        rs = gemm_tn_ring_splits(M, N, K)
        if rs:                                                   # the ring kernel's stage-0 geometry: the whole [N, K] gradient per workgroup
            part, ch = gemm_tn_ring(dy, x, rs, torch.bfloat16 if (GEMM_TN_RING_BF16 and rs > 1) else torch.float32), rs
            if ch == 1:
                dw = part.view(N, K).float()
        else:
            ch = _pick_split(M, -(-N // 64) * -(-K // 64))
            with _lib.timed("lib_gemm_wgrad", 2 * (M * K + M * N) + 4 * N * K, 2 * M * K * N):
                if ch > 1:
                    part = torch.bmm(dy.view(ch, M // ch, N).transpose(1, 2), x.view(ch, M // ch, K))
                else:
                    dw = (dy.t() @ x).float()
        if ch > 1:
            dw = sum_rows(part, ch, N * K, out=grad_slot(ctx.weight), owners=(ctx.weight,)).view(N, K)
        else:
            flush_if_pending((ctx.weight,))
        return dx, dw, db, None


class _Mlp0(torch.autograd.Function):
    """fc2_nobias(gelu(fc1(x))) of the stage-0 Mlp (HOT:50-58, C = 96) as one autograd node.  Forward: pswin_fc1_gelu_fwd (h is kept,
    the pre-activation is not) + the streaming GEMM.  Backward: g = (dy W2) * gelu'(x W1^T + b1) in ONE pass (pswin_mlp0_bwd: fc2's
    data gradient never reaches memory), then dx = g W1 (streaming GEMM), dW1 = g^T x and dW2 = dy^T h (ring kernel)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, w1_lp, w2_lp):
        x = x.contiguous()
        w1b = (w1_lp if w1_lp is not None else w1.to(x.dtype)).contiguous()
        w2b = (w2_lp if w2_lp is not None else w2.to(x.dtype)).contiguous()
        M, K = x.shape
        N = w1b.shape[0]
        h = torch.empty(M, N, dtype=x.dtype, device=x.device)
        b = b1.detach().float().contiguous()
        if MLP0_FUSED_FWD:                                   # both products in one pass: h is written once and not read back
            y = torch.empty(M, K, dtype=x.dtype, device=x.device)
            call("pswin_mlp0_fwd", x, ptr(x), ptr(w1b), ptr(b), ptr(w2b), ptr(h), ptr(y), M, K, N,
                 algo_bytes=2 * M * (2 * K + N), algo_flops=4 * M * K * N)
        else:
            call("pswin_fc1_gelu_fwd", x, ptr(x), ptr(w1b), ptr(b), ptr(h), M, K, N, algo_bytes=2 * M * (K + N))
            y = skinny_gemm(h, w2b, None)
        ctx.save_for_backward(x, w1b, b, h, w2b)
        ctx.params = (w1, b1, w2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1b, b, h, w2b = ctx.saved_tensors
        w1, b1, w2 = ctx.params
        M, K = x.shape
        N = w1b.shape[0]
        dy = dy.to(x.dtype).contiguous()
        g = torch.empty(M, N, dtype=x.dtype, device=x.device)
        lib = _lib.load()
        ring_bias = GEMM_TN_RING_BIAS and gemm_tn_ring_splits(M, N, K) > 0     # the fc1 weight-gradient launch also sums g's columns
        ws = None
        if not ring_bias:
            rows = lib.pswin_mlp0_bwd_partial_rows(M)
            ws = torch.empty(rows, N, dtype=torch.float32, device=x.device)
        call("pswin_mlp0_bwd", x, ptr(x), ptr(w1b), ptr(b), ptr(dy), ptr(w2b), ptr(g), None, ptr(ws), M, K, N,
             algo_bytes=2 * M * (2 * K + N), algo_flops=4 * M * K * N)
        if ring_bias:
            dx, dw1, db1 = linear_backward(x, w1b, g, w1, b1, None, ctx.needs_input_grad[0])
        else:
            db1 = sum_rows(ws, rows, N, owners=(b1,))
            dx, dw1, _ = linear_backward(x, w1b, g, w1, None, None, ctx.needs_input_grad[0])
        _, dw2, _ = linear_backward(h, w2b, dy, w2, None, None, False)
        return dx, dw1, db1, dw2, None, None


# off: the stage-0 Mlp as fc1 + GELU node and fc2 node (pswin_fc1_gelu_bwd)
MLP0_FUSED = _on("mlp0_fused")
# off: its forward as pswin_fc1_gelu_fwd + the streaming GEMM for fc2
MLP0_FUSED_FWD = _on("mlp0_fused_fwd")


def mlp0_fused_shape_ok(M, C, hidden):
    """pswin_mlp0_fwd / _bwd take [M, C] rows and an [M, hidden] activation: both must be 32-bit addressable (M < ~5.6 M rows at
    hidden = 384: batch 170 at 512 x 1024, batch 42 at 1024 x 2048); larger batches fall through to the unfused path."""
    lib = _lib.load()
    return (M >= 4096 and rows_addressable(M, max(C, hidden)) and bool(lib.pswin_mlp0_bwd_supported(C, hidden))
            and fc1_gelu_shape_ok(M, C, hidden) and skinny_gemm_shape_ok(M, hidden, C))


def mlp0_fused_supported(x2d, hidden):
    return (MLP0_FUSED and x2d.dtype == torch.bfloat16 and x2d.is_cuda and x2d.dim() == 2
            and mlp0_fused_shape_ok(x2d.shape[0], x2d.shape[1], hidden))


def mlp0_fused(x2d, fc1, fc2):
    """fc2_nobias(gelu(fc1(x2d))) for the stage-0 Mlp as one autograd node: see _Mlp0."""
    l1, l2 = fc1.__dict__.get("_lowp"), fc2.__dict__.get("_lowp")
    return _Mlp0.apply(x2d, fc1.weight, fc1.bias, fc2.weight, l1[0] if l1 is not None else None, l2[0] if l2 is not None else None)


def fc1_gelu_shape_ok(M, K, N):
    return M >= 4096 and rows_addressable(M, max(K, N)) and bool(_lib.load().pswin_fc1_gelu_supported(K, N))


def fc1_gelu_supported(x2d, n_out):
    return x2d.dtype == torch.bfloat16 and x2d.is_cuda and fc1_gelu_shape_ok(x2d.shape[0], x2d.shape[1], n_out)


def fc1_gelu(x2d, weight, bias, w_lp=None):
    return _Fc1Gelu.apply(x2d, weight, bias, w_lp)


class _BiasGelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, bias):
        y = y.contiguous()
        M, N = y.numel() // y.shape[-1], y.shape[-1]
        h = torch.empty_like(y)
        b = None if bias is None else bias.detach().float().contiguous()
        call("pswin_bias_gelu_fwd", y, ptr(y), dtype_code(y), ptr(b), ptr(h), M, N, algo_bytes=2 * y.numel() * y.element_size())
        ctx.save_for_backward(y, b)
        ctx.bias = bias
        return h

    @staticmethod
    def backward(ctx, dh):
        y, b = ctx.saved_tensors
        M, N = y.numel() // y.shape[-1], y.shape[-1]
        dh = dh.to(y.dtype).contiguous()
        dy = torch.empty_like(y)
        lib = _lib.load()
        ws = torch.empty(lib.pswin_bias_gelu_workspace(M, N), dtype=torch.float32, device=y.device)
        call("pswin_bias_gelu_bwd", y, ptr(dh), ptr(y), dtype_code(y), ptr(b), ptr(dy), None, ptr(ws), M, N,
             algo_bytes=3 * y.numel() * y.element_size())
        db = sum_rows(ws, lib.pswin_bias_gelu_partial_rows(M, N, dtype_code(y)), N, owners=(ctx.bias,)) \
            if b is not None else None
        return dy, db


class _BiasGeluLinear(torch.autograd.Function):
    """out = gelu(y + b1) @ W2^T: fc1's bias + nn.GELU followed by fc2 WITHOUT its bias (HOT:50-58; the caller adds fc2's bias with
    the residual).  One function so that the backward pass can run the data gradient of fc2 and the backward of bias + GELU
    in ONE kernel (pswin_gemm_nt_gelu_bwd): dL/dh [M, 4C] is never written or re-read, and the fc1 bias gradient comes
    out of the same epilogue as per-tile column sums."""

    @staticmethod
    def forward(ctx, y, b1, w2, w2_lp, w2_lp_t):
        y = y.contiguous()
        shp = y.shape
        y2 = y.view(-1, shp[-1])
        M, N = y2.shape
        h = torch.empty_like(y2)
        b = None if b1 is None else b1.detach().float().contiguous()
        call("pswin_bias_gelu_fwd", y2, ptr(y2), dtype_code(y2), ptr(b), ptr(h), M, N, algo_bytes=2 * y2.numel() * y2.element_size())
        wb = w2_lp if w2_lp is not None else w2.to(y.dtype)
        C = wb.shape[0]
        tile = gemm_nt_tile(M, N, C)
        if tile:
            out = gemm_nt(h, wb, None, tile)
        else:
            with _lib.timed("lib_gemm_fwd", 2 * (M * N + M * C + N * C), 2 * M * N * C):
                out = F.linear(h, wb)
        ctx.save_for_backward(y2, b, h, wb, w2_lp_t)
        ctx.params = (b1, w2)
        return out.view(*shp[:-1], C)

    @staticmethod
    def backward(ctx, dout):
        y2, b, h, wb, wbt = ctx.saved_tensors
        b1, w2 = ctx.params
        M, N = y2.shape
        C = wb.shape[0]
        dout = dout.reshape(M, C).contiguous()
        lib = _lib.load()
        tile = 0
        if wbt is not None and GEMM_NT and lib.pswin_gemm_nt_supported(M, C, N):
            tile = gemm_nt_tile(M, C, N) or gemm_nt_rows(M, N)        # (the narrow stage-3 case the plain rule leaves to the library: fused it wins)
        if tile:
            dpre = torch.empty_like(y2)
            rows = lib.pswin_gemm_nt_partial_rows(M, tile)
            ws = torch.empty(rows, N, dtype=torch.float32, device=y2.device)
            call("pswin_gemm_nt_gelu_bwd", y2, ptr(dout), ptr(wbt), ptr(y2), ptr(b), ptr(dpre), ptr(ws), M, C, N, tile,
                 algo_bytes=2 * (M * C + 2 * M * N + N * C), algo_flops=2 * M * N * C)
            db = sum_rows(ws, rows, N, owners=(b1,)) if b is not None else None
        else:
            with _lib.timed("lib_gemm_dgrad", 2 * (M * N + M * C + N * C), 2 * M * N * C):
                dh = dout @ wb
            dpre = torch.empty_like(y2)
            ws = torch.empty(lib.pswin_bias_gelu_workspace(M, N), dtype=torch.float32, device=y2.device)
            call("pswin_bias_gelu_bwd", y2, ptr(dh), ptr(y2), dtype_code(y2), ptr(b), ptr(dpre), None, ptr(ws), M, N,
                 algo_bytes=3 * y2.numel() * y2.element_size())
            db = sum_rows(ws, lib.pswin_bias_gelu_partial_rows(M, N, dtype_code(y2)), N, owners=(b1,)) if b is not None else None
        _, dw, _ = linear_backward(h, wb, dout, w2, None, None, False)
        return dpre.view_as(y2), db, dw, None, None


class _MlpFused(torch.autograd.Function):
    """fc2(gelu(fc1(x) + b1)) WITHOUT fc2's bias for bf16 rows on the tiled GEMM (stages 1-3; HOT:44-61): fc1 with the bias + GELU
    in its epilogue (pswin_gemm_nt_gelu_fwd: the pre-activation is written once and never re-read in the forward pass), fc2;
    backward: fc2's data gradient with the GELU backward and the fc1 bias gradient in its epilogue (pswin_gemm_nt_gelu_bwd),
    fc1's data gradient on the transposed weight copy, the two weight gradients as everywhere."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, w1_lp, w1_lp_t, w2_lp, w2_lp_t):
        x = x.contiguous()
        M, K = x.shape
        w1b = w1_lp if w1_lp is not None else w1.to(x.dtype)
        w2b = w2_lp if w2_lp is not None else w2.to(x.dtype)
        N, C = w1b.shape[0], w2b.shape[0]
        b = b1.detach().float().contiguous()
        pre = torch.empty(M, N, dtype=x.dtype, device=x.device)
        h = torch.empty_like(pre)
        call("pswin_gemm_nt_gelu_fwd", x, ptr(x), ptr(w1b), ptr(b), ptr(pre), ptr(h), M, K, N, gemm_nt_tile(M, K, N) or gemm_nt_rows(M, N),
             algo_bytes=2 * (M * K + 2 * M * N + N * K), algo_flops=2 * M * K * N)
        tile = gemm_nt_tile(M, N, C)
        if tile:
            out = gemm_nt(h, w2b, None, tile)
        else:
            with _lib.timed("lib_gemm_fwd", 2 * (M * N + M * C + N * C), 2 * M * N * C):
                out = F.linear(h, w2b)
        ctx.save_for_backward(x, pre, h, b, w1b, w1_lp_t, w2b, w2_lp_t)
        ctx.params = (w1, b1, w2)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, pre, h, b, w1b, w1bt, w2b, w2bt = ctx.saved_tensors
        w1, b1, w2 = ctx.params
        M, N = pre.shape
        C = w2b.shape[0]
        dout = dout.contiguous()
        lib = _lib.load()
        if w2bt is not None:
            tile = gemm_nt_tile(M, C, N) or gemm_nt_rows(M, N)
            dpre = torch.empty_like(pre)
            rows = lib.pswin_gemm_nt_partial_rows(M, tile)
            ws = torch.empty(rows, N, dtype=torch.float32, device=pre.device)
            call("pswin_gemm_nt_gelu_bwd", pre, ptr(dout), ptr(w2bt), ptr(pre), ptr(b), ptr(dpre), ptr(ws), M, C, N, tile,
                 algo_bytes=2 * (M * C + 2 * M * N + N * C), algo_flops=2 * M * N * C)
            db = sum_rows(ws, rows, N, owners=(b1,))
        else:
            with _lib.timed("lib_gemm_dgrad", 2 * (M * N + M * C + N * C), 2 * M * N * C):
                dh = dout @ w2b
            dpre = torch.empty_like(pre)
            ws = torch.empty(lib.pswin_bias_gelu_workspace(M, N), dtype=torch.float32, device=pre.device)
            call("pswin_bias_gelu_bwd", pre, ptr(dh), ptr(pre), dtype_code(pre), ptr(b), ptr(dpre), None, ptr(ws), M, N,
                 algo_bytes=3 * pre.numel() * pre.element_size())
            db = sum_rows(ws, lib.pswin_bias_gelu_partial_rows(M, N, dtype_code(pre)), N, owners=(b1,))
        _, dw2, _ = linear_backward(h, w2b, dout, w2, None, None, False)
        dx, dw1, _ = linear_backward(x, w1b, dpre, w1, None, None, ctx.needs_input_grad[0], w1bt)
        return dx, dw1, db, dw2, None, None, None, None


def mlp_fused_supported(x2d, hidden):
    return (GEMM_NT and FUSED_GELU_BWD and FUSED_MLP and x2d.dtype == torch.bfloat16 and x2d.is_cuda and x2d.dim() == 2 and x2d.shape[0] >= 64
            and bool(_lib.load().pswin_gemm_nt_supported(x2d.shape[0], x2d.shape[1], hidden)))


def mlp_fused(x2d, fc1, fc2):
    """fc2_nobias(gelu(fc1(x2d))) as one autograd node on the tiled GEMM kernels: see _MlpFused."""
    l1, l2 = fc1.__dict__.get("_lowp"), fc2.__dict__.get("_lowp")
    return _MlpFused.apply(x2d, fc1.weight, fc1.bias, fc2.weight, l1[0] if l1 is not None else None, fc1.__dict__.get("_lowp_t"),
                           l2[0] if l2 is not None else None, fc2.__dict__.get("_lowp_t"))


def bias_gelu_linear(y, bias1, lin2):
    """lin2(gelu(y + bias1)) without lin2's bias (bf16 rows): see _BiasGeluLinear."""
    lp = lin2.__dict__.get("_lowp")
    out = _BiasGeluLinear.apply(y.reshape(-1, y.shape[-1]), bias1, lin2.weight, lp[0] if lp is not None else None,
                                lin2.__dict__.get("_lowp_t"))
    return out.view(*y.shape[:-1], lin2.weight.shape[0])


def bias_gelu(y, bias):
    """gelu(y + bias) (exact, erf) on [..., N]; the backward pass also returns the bias gradient (column sums of
    dy) from the same pass: fc1 bias + nn.GELU of Mlp (HOT:50-57) without a GEMM epilogue or a separate reduction."""
    return _BiasGelu.apply(y, bias)


class _InterpRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx, wgt):
        B, S, C = x.shape
        P = idx.shape[0]
        ctx.save_for_backward(idx, wgt)
        ctx.geom = (B, S, P, C)
        x = x.contiguous()
        out = torch.empty(B, P, C, dtype=torch.float32, device=x.device)
        call("pswin_interp_rows", x, ptr(x), ptr(idx), ptr(wgt), ptr(out), B, S, P, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        idx, wgt = ctx.saved_tensors
        B, S, P, C = ctx.geom
        dout = dout.contiguous()
        dx = torch.zeros(B, S, C, dtype=torch.float32, device=dout.device)
        call("pswin_interp_rows_adjoint", dout, ptr(dout), ptr(idx), ptr(wgt), ptr(dx), B, S, P, C)
        return dx, None, None


def interp_rows(x, idx, wgt):
    """out[b, p] = sum_k wgt[p, k] x[b, idx[p, k]]: a static bilinear F.grid_sample as a 4-tap row gather."""
    return _InterpRows.apply(x.float(), idx, wgt)


# ------------------------------------------------------------------------------------------------
# window attention
# ------------------------------------------------------------------------------------------------
class Tiles:
    """A [n, 49, 49] fp32 table as zero-padded 64 x 64 tiles: `fwd` holds tile[i][j], `bwd` the transposed tile[j][i]
    (the same buffer when the table is symmetric, e.g. self-attention distances and the shifted-window mask)."""

    def __init__(self, table, symmetric=False):
        table = table.contiguous().float()
        assert table.dim() == 3 and table.shape[1:] == (WTOK, WTOK)
        self.n = table.shape[0]
        self.table = table
        self.fwd = torch.empty(self.n, WPAD, WPAD, dtype=torch.float32, device=table.device)
        call("pswin_attn_pad_tiles", table, ptr(table), self.n, 0, ptr(self.fwd))
        if symmetric:
            self.bwd = self.fwd
        else:
            self.bwd = torch.empty_like(self.fwd)
            call("pswin_attn_pad_tiles", table, ptr(table), self.n, 1, ptr(self.bwd))


def window_dist_tiles(H, W, shift, device):
    """Tiles of window_dist (symmetric: haversine of a window with itself)."""
    key = ("dist_tiles", H, W, shift, _dev_key(device))
    if key not in _CACHE:
        _CACHE[key] = Tiles(window_dist(H, W, shift, device), symmetric=True)
    return _CACHE[key]


def planar_mask_tiles(H, W, shift, device):
    key = ("mask_tiles", H, W, shift, _dev_key(device))
    if key not in _CACHE:
        _CACHE[key] = Tiles(planar_mask(H, W, shift, device), symmetric=True)
    return _CACHE[key]


def _as_tiles(t):
    if t is None or isinstance(t, Tiles):
        return t
    return Tiles(t)


class _WindowAttention(torch.autograd.Function):
    """softmax(scale q k^T + bias) v per (window, head); q, k, v either slices of one fused [rows, 3C] buffer
    (k = v = None) or three separate [rows, C] tensors."""

    @staticmethod
    def forward(ctx, q_or_qkv, k, v, alpha, beta, dist, mask, heads, scale, n_bias_windows, chunks):
        import ctypes
        fused = k is None
        x = q_or_qkv.contiguous()
        C = heads * _lib.HEAD_DIM
        rows = x.shape[0]
        assert rows % WTOK == 0
        n = rows // WTOK
        if fused:
            assert x.shape[1] == 3 * C, "fused qkv must be [rows, 3C] with C = heads * 32"
            ld = 3 * C
            es = x.element_size()
            qp, kp, vp = x.data_ptr(), x.data_ptr() + C * es, x.data_ptr() + 2 * C * es
        else:
            k, v = k.contiguous(), v.contiguous()
            assert x.shape == k.shape == v.shape and x.shape[1] == C
            assert x.dtype == k.dtype == v.dtype
            ld = C
            qp, kp, vp = x.data_ptr(), k.data_ptr(), v.data_ptr()
        alpha_p, beta_p = (alpha if dist is not None else None), beta
        alpha = alpha.contiguous() if dist is not None else None
        beta = beta.contiguous()
        out = torch.empty(rows, C, dtype=x.dtype, device=x.device)
        lse = torch.empty(n, heads, WPAD, dtype=torch.float32, device=x.device)
        call("pswin_attn_fwd", x, ctypes.c_void_p(qp), ctypes.c_void_p(kp), ctypes.c_void_p(vp), ld,
             ptr(None if dist is None else dist.fwd), 0 if dist is None else dist.n, ptr(alpha), ptr(beta),
             ptr(None if mask is None else mask.fwd), 0 if mask is None else mask.n, ptr(out), C, ptr(lse),
             int(chunks or 0), n, n_bias_windows, heads, float(scale), dtype_code(x),
             algo_bytes=n * heads * 4 * WTOK * _lib.HEAD_DIM * x.element_size())
        ctx.fused, ctx.heads, ctx.scale, ctx.nb, ctx.n, ctx.C = fused, heads, float(scale), n_bias_windows, n, C
        ctx.dist, ctx.mask, ctx.chunks = dist, mask, chunks
        ctx.owners = (alpha_p, beta_p)
        ctx.save_for_backward(x, k, v, lse, alpha, beta)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, k, v, lse, alpha, beta = ctx.saved_tensors
        dx, dk, dv, dalpha, dbeta = attention_backward(x, k, v, lse, alpha, beta, ctx.dist, ctx.mask, dout, ctx.heads, ctx.scale,
                                                       ctx.nb, ctx.chunks, ctx.needs_input_grad[3] or ctx.needs_input_grad[4],
                                                       ctx.owners)
        return dx, dk, dv, dalpha, dbeta, None, None, None, None, None, None


def attention_backward(x, k, v, lse, alpha, beta, dist, mask, dout, heads, scale, nb, chunks, need_tables, owners, packed=False):
    """Backward of the attention core (pswin_attn_bwd + the table-gradient kernels): x = fused [rows, 3C] qkv (k = v =
    None) or q with separate k, v; packed=True: x = [n, heads, 3, 49, 32] (what the fused forward kernel saves), the
    gradient still comes back as one row-major [rows, 3C] tensor.  Returns (dx, dk, dv, dalpha, dbeta); owners = the
    (alpha, beta) parameters."""
    import ctypes
    C = heads * _lib.HEAD_DIM
    n = x.shape[0] if packed else x.shape[0] // WTOK
    fused = k is None
    dout = dout.contiguous()
    es = x.element_size()
    ld_in, win_stride, head_stride = None, None, _lib.HEAD_DIM
    if packed:
        dx = torch.empty(n * WTOK, 3 * C, dtype=x.dtype, device=x.device)
        ld = 3 * C
        blk = WTOK * _lib.HEAD_DIM
        qp, kp, vp = x.data_ptr(), x.data_ptr() + blk * es, x.data_ptr() + 2 * blk * es
        dqp, dkp, dvp = dx.data_ptr(), dx.data_ptr() + C * es, dx.data_ptr() + 2 * C * es
        dk = dv = None
        ld_in, win_stride, head_stride = _lib.HEAD_DIM, heads * 3 * blk, 3 * blk
    elif fused:
        dx = torch.empty_like(x)
        ld = 3 * C
        qp, kp, vp = x.data_ptr(), x.data_ptr() + C * es, x.data_ptr() + 2 * C * es
        dqp, dkp, dvp = dx.data_ptr(), dx.data_ptr() + C * es, dx.data_ptr() + 2 * C * es
        dk = dv = None
    else:
        dx, dk, dv = torch.empty_like(x), torch.empty_like(k), torch.empty_like(v)
        ld = C
        qp, kp, vp = x.data_ptr(), k.data_ptr(), v.data_ptr()
        dqp, dkp, dvp = dx.data_ptr(), dk.data_ptr(), dv.data_ptr()
    lib = _lib.load()
    chunks = chunks or lib.pswin_attn_suggest_chunks(n, nb, heads, 1)
    dalpha = dbeta = gsum = None
    if need_tables:
        gsum = torch.empty(chunks * nb, heads, WPAD, WPAD, dtype=torch.float32, device=x.device)
    ld_in = ld if ld_in is None else ld_in
    win_stride = WTOK * ld_in if win_stride is None else win_stride
    call("pswin_attn_bwd_ex", x, ctypes.c_void_p(qp), ctypes.c_void_p(kp), ctypes.c_void_p(vp), ld_in, win_stride, head_stride,
         ptr(None if dist is None else dist.bwd), 0 if dist is None else dist.n, ptr(alpha), ptr(beta),
         ptr(None if mask is None else mask.bwd), 0 if mask is None else mask.n, ptr(dout), C, ptr(lse),
         ctypes.c_void_p(dqp), ctypes.c_void_p(dkp), ctypes.c_void_p(dvp), ld, ptr(gsum),
         chunks, n, nb, heads, scale, dtype_code(x),
         algo_bytes=n * heads * 7 * WTOK * _lib.HEAD_DIM * x.element_size())
    if need_tables:
        dbeta = torch.empty(169, heads, dtype=torch.float32, device=x.device)
        dalpha = torch.empty_like(dbeta) if dist is not None else None
        ws = torch.empty(lib.pswin_attn_table_grads_workspace(heads), dtype=torch.float32, device=x.device)
        job = (gsum, None if dist is None else dist.bwd, dalpha, dbeta, ws, chunks * nb, nb,
               0 if dist is None else dist.n, heads)
        # with deferred reductions the whole table gradient waits for the end of the pass: ONE launch sums the dScore tiles of all
        # attention modules (round 4: twelve launches of 9-15 us per step were mostly ramp; the tiles -- 16 KB per work item and head,
        # ~190 MB per PanoSwin-T step at batch 8 -- stay referenced by the queue until then), the sum of its partial rows joins the grouped
        # reduction and ONE binning launch follows (same kernels, same per-module decomposition either way: bitwise equal results)
        ld = ws.numel() // 129
        rjob = (ws, 0, F32, lib.pswin_attn_table_grads_partial_rows(chunks * nb, heads), ld, ld, ws[128 * ld:])
        q = _deferring(owners)
        if q is None or not DEFER_TABLE_PARTIALS:
            _launch_table_grads([job], 1)
        if q is not None:
            q["jobs"].append(rjob)
            q["table_jobs"].append(job)
            dbeta = dbeta.view(169, heads)                       # fresh views: see sum_rows
            dalpha = None if dalpha is None else dalpha.view(169, heads)
        else:
            _launch_reductions([rjob])
            _launch_table_grads([job], 4)
    return dx, dk, dv, dalpha, dbeta


def window_attention(qkv, alpha, beta, dist, mask, heads, scale, n_bias_windows, k=None, v=None, chunks=None):
    """BasicWindowAttention.forward between qkv and proj (HOT:288-308).

    qkv: [n*49, 3C] (or q with k, v given: three [n*49, C]); alpha/beta: [169, heads] f32 tables;
    dist: great-circle table, a [nW, 49, 49] f32 tensor or a prebuilt ``Tiles`` (None in planar mode: beta only,
    HOT:257-258); mask: f32 [nM, 49, 49] tensor / ``Tiles`` / None; window n uses bias window n % n_bias_windows.
    chunks: work items per bias window (None = library heuristic).  Returns [n*49, C].
    """
    return _WindowAttention.apply(qkv, k, v, alpha, beta, _as_tiles(dist), _as_tiles(mask), heads, scale,
                                  n_bias_windows, chunks)


class _WindowAttentionFused(torch.autograd.Function):
    """WindowAttention.forward (HOT:274-323) as ONE kernel per direction-of-use: qkv Linear, attention core and proj Linear
    of a window run out of registers (pswin_win_attn_fused_fwd); the qkv tensor is written only when a backward pass
    will read it.  The backward pass is the unfused chain on the saved tensors (proj gradients, pswin_attn_bwd, qkv
    gradients): identical kernels and summation order to the unfused path."""

    @staticmethod
    def forward(ctx, x, w_qkv, b_qkv, w_proj, alpha, beta, dist, mask, heads, scale, n_bias_windows, wq_lp, wp_lp, grad_mode):
        x = x.contiguous()
        rows, C = x.shape
        n = rows // WTOK
        assert rows % WTOK == 0 and C == heads * _lib.HEAD_DIM
        wq = (wq_lp if wq_lp is not None else w_qkv.to(x.dtype)).contiguous()
        wp = (wp_lp if wp_lp is not None else w_proj.to(x.dtype)).contiguous()
        bq = None if b_qkv is None else b_qkv.detach().float().contiguous()
        alpha_c = alpha.detach().contiguous() if dist is not None else None
        beta_c = beta.detach().contiguous()
        # grad_mode = torch.is_grad_enabled() of the CALLER (inside forward() it is always off, and needs_input_grad
        # ignores torch.no_grad()): without a backward pass to come, qkv / the attention output / lse are never written
        save = grad_mode and any(ctx.needs_input_grad)
        y = torch.empty_like(x)
        qkv = att = lse = None
        if save:
            qkv = torch.empty(n, heads, 3, WTOK, _lib.HEAD_DIM, dtype=x.dtype, device=x.device)     # packed blocks (pswin_attn_bwd_ex)
            att = torch.empty_like(x)
            lse = torch.empty(n, heads, WPAD, dtype=torch.float32, device=x.device)
        flops = n * (2 * WTOK * C * 3 * C + heads * 4 * WTOK * WTOK * _lib.HEAD_DIM + 2 * WTOK * C * C)
        call("pswin_win_attn_fused_fwd", x, ptr(x), ptr(wq), ptr(bq), ptr(wp), ptr(None if dist is None else dist.fwd),
             0 if dist is None else dist.n, ptr(alpha_c), ptr(beta_c), ptr(None if mask is None else mask.fwd),
             0 if mask is None else mask.n, ptr(y), ptr(qkv), ptr(att), ptr(lse), n, n_bias_windows, C, heads, float(scale),
             dtype_code(x), algo_bytes=rows * C * 2 * (6 if save else 2), algo_flops=flops)
        if save:
            ctx.save_for_backward(x, qkv, att, lse, wq, wp, alpha_c, beta_c)
            ctx.params = (w_qkv, b_qkv, w_proj, alpha if dist is not None else None, beta)
            ctx.cfg = (dist, mask, heads, float(scale), n_bias_windows)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, qkv, att, lse, wq, wp, alpha_c, beta_c = ctx.saved_tensors
        w_qkv, b_qkv, w_proj, alpha, beta = ctx.params
        dist, mask, heads, scale, nb = ctx.cfg
        C = x.shape[1]
        datt, dwp, _ = linear_backward(att, wp, dy, w_proj, None, None, True)
        dqkv, _, _, dalpha, dbeta = attention_backward(qkv, None, None, lse, alpha_c, beta_c, dist, mask, datt, heads, scale, nb,
                                                        None, ctx.needs_input_grad[4] or ctx.needs_input_grad[5], (alpha, beta),
                                                        packed=True)
        # the K third of d(qkv) sums to zero over every window (rows of dS sum to 0): its bias gradient is not summed
        dx, dwq, dbq = linear_backward(x, wq, dqkv, w_qkv, b_qkv, (C, 2 * C), ctx.needs_input_grad[0])
        return dx, dwq, dbq, dwp, dalpha, dbeta, None, None, None, None, None, None, None, None


def fused_windows_addressable(n_windows, C):
    """the bound the fused attention launchers enforce on the packed q, k, v save: n_windows * 49 * 3C * 2 bytes < 0x7fffffff00"""
    return n_windows * WTOK * 3 * C * 2 < 0x7fffffff00


def window_attention_fused_supported(x2d, heads):
    return (x2d.is_cuda and x2d.dim() == 2 and x2d.dtype == torch.bfloat16 and fused_windows_addressable(x2d.shape[0] // WTOK, x2d.shape[1])
            and bool(_lib.load().pswin_win_attn_fused_supported(x2d.shape[1], heads, BF16)))


def window_attention_fused(x2d, attn, dist, mask, n_bias_windows):
    """proj(attention(qkv(x2d))) WITHOUT the proj bias for window rows x2d [n*49, C] and the parameter holder `attn`
    (qkv, proj, the two tables, num_heads, scale): see _WindowAttentionFused."""
    lq, lp = attn.qkv.__dict__.get("_lowp"), attn.proj.__dict__.get("_lowp")
    return _WindowAttentionFused.apply(x2d, attn.qkv.weight, attn.qkv.bias, attn.proj.weight,
                                       attn.sphere_position_alpha_table_Te, attn.sphere_position_beta_table_Te,
                                       _as_tiles(dist), _as_tiles(mask), attn.num_heads, attn.scale, n_bias_windows,
                                       lq[0] if lq is not None else None, lp[0] if lp is not None else None,
                                       torch.is_grad_enabled())


# ------------------------------------------------------------------------------------------------
# the caller's side: RoIAlign over the FPN pyramid
# ------------------------------------------------------------------------------------------------
def _roi_levels(ptrs_fwd, ptrs_bwd, shapes, strides):
    lv = _lib.RoiLevels()
    lv.n_levels = len(shapes)
    for l, ((H, W), s) in enumerate(zip(shapes, strides)):
        lv.feat[l] = ptrs_fwd[l] if ptrs_fwd else None
        lv.dfeat[l] = ptrs_bwd[l] if ptrs_bwd else None
        lv.H[l], lv.W[l], lv.spatial_scale[l] = H, W, 1.0 / s
    return lv


class _RoiAlignFPN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rois, roi_level, P, sampling_ratio, aligned, strides, *feats):
        import ctypes
        C = feats[0].shape[1]
        dt = feats[0].dtype
        if len(feats) > 4 or any(f.dtype != dt or f.shape[1] != C or f.dim() != 4 for f in feats):
            raise PswinError("roi_align_fpn: up to four NCHW maps of one dtype and channel count")
        if not _lib.load().pswin_roi_align_supported(C, dtype_code(feats[0])):
            raise PswinError(f"roi_align_fpn: unsupported channel count {C} for {dt}")
        nhwc = [f.permute(0, 2, 3, 1).contiguous() for f in feats]          # a pixel's channels = one contiguous row
        rois = rois.detach().float().contiguous()
        roi_level = roi_level.to(torch.int32).contiguous()
        R = rois.shape[0]
        out = torch.empty(R, P, P, C, dtype=dt, device=rois.device)
        shapes = [tuple(f.shape[2:]) for f in feats]
        lv = _roi_levels([f.data_ptr() for f in nhwc], None, shapes, strides)
        call("pswin_roi_align_fwd", out, ctypes.byref(lv), ptr(rois), ptr(roi_level), R, C, P, int(sampling_ratio), int(aligned),
             dtype_code(out), ptr(out))
        ctx.save_for_backward(rois, roi_level)
        ctx.cfg = (P, int(sampling_ratio), int(aligned), tuple(strides), shapes, [f.shape[0] for f in feats], C, dt)
        return out.permute(0, 3, 1, 2)                                      # [R, C, P, P] (channels-last in memory)

    @staticmethod
    def backward(ctx, dout):
        import ctypes
        rois, roi_level = ctx.saved_tensors
        P, sr, aligned, strides, shapes, batches, C, dt = ctx.cfg
        d = dout.permute(0, 2, 3, 1).to(dt).contiguous()                    # [R, P, P, C]
        grads = [torch.zeros(b, H, W, C, dtype=torch.float32, device=d.device) for b, (H, W) in zip(batches, shapes)]
        lv = _roi_levels(None, [g.data_ptr() for g in grads], shapes, strides)
        call("pswin_roi_align_bwd", d, ctypes.byref(lv), ptr(rois), ptr(roi_level), rois.shape[0], C, P, sr, aligned, dtype_code(d), ptr(d))
        return (None, None, None, None, None, None) + tuple(g.permute(0, 3, 1, 2).to(dt) for g in grads)


def map_roi_levels(rois, num_levels, finest_scale=56):
    """SingleRoIExtractor.map_roi_levels (mmdet/models/roi_heads/roi_extractors/single_level_roi_extractor.py:55-60): rois [R, 5]."""
    scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
    return torch.floor(torch.log2(scale / finest_scale + 1e-6)).clamp(min=0, max=num_levels - 1).long()


def roi_align_fpn(feats, strides, rois, out_size, sampling_ratio=0, aligned=True, finest_scale=56, roi_level=None):
    """SingleRoIExtractor(RoIAlign(out_size, sampling_ratio)) over up to four NCHW pyramid levels (single_level_roi_extractor.py:78-108,
    configs/_base_/models/mask_rcnn_swin_fpn.py:44-48): rois [R, 5] = (batch index, x1, y1, x2, y2) in image pixels -> [R, C, out, out].
    Forward and backward are HIP kernels (pswin_roi_align_fwd / _bwd); only the feature maps receive a gradient."""
    if roi_level is None:
        roi_level = map_roi_levels(rois, len(feats), finest_scale)
    return _RoiAlignFPN.apply(rois, roi_level, int(out_size), sampling_ratio, aligned, tuple(strides), *feats)


_NMS_COUNTS = {}


def nms_groups(box_list, iou_thr):
    """Greedy NMS of several score-sorted box lists in ONE launch (pswin_nms_groups; a workgroup per list): list of bool keep masks.
    box_list: f32 [n_g, 4] tensors on one HIP device, n_g <= 2048 (the RPN's nms_pre = 2000 per image and pyramid level)."""
    import ctypes
    ns = tuple(int(b.shape[0]) for b in box_list)
    dev = box_list[0].device
    nmax = -(-max(ns) // 64) * 64                    # the kernels work on whole 64-row chunks
    if nmax == 0:
        return [torch.zeros(0, dtype=torch.bool, device=dev) for _ in ns]
    if nmax > 2048:
        raise PswinError(f"nms_groups: at most 2048 boxes per list, got {max(ns)}")
    key = (ns, _dev_key(dev))
    if key not in _NMS_COUNTS:                      # static per feature-map geometry: no host-to-device copy inside a captured step
        _NMS_COUNTS[key] = torch.tensor(ns, dtype=torch.int32, device=dev)
    boxes = torch.zeros(len(ns), nmax, 4, dtype=torch.float32, device=dev)
    for g, b in enumerate(box_list):
        boxes[g, :ns[g]] = b.detach().float()
    keep = torch.empty(len(ns), nmax, dtype=torch.uint8, device=dev)
    ws = torch.empty(int(_lib.load().pswin_nms_workspace(len(ns), nmax)), dtype=torch.uint8, device=dev)
    call("pswin_nms_groups", boxes, ptr(boxes), ptr(_NMS_COUNTS[key]), len(ns), nmax, ctypes.c_float(float(iou_thr)), ptr(keep), ptr(ws))
    return [keep[g, :ns[g]].bool() for g in range(len(ns))]


# ------------------------------------------------------------------------------------------------
# qkv Linear + attention core in one kernel for C = 192 / 384 (pswin_qkv_attn_fused_fwd, round 3)
# ------------------------------------------------------------------------------------------------
# off: the unfused chain qkv GEMM -> pswin_attn_fwd
FUSED_QKV_ATTENTION = _on("fused_qkv_attention")


class _WindowAttentionQkvFused(torch.autograd.Function):
    """self.qkv(x) and the attention core of WindowAttention.forward (HOT:287-308) as ONE kernel per (window, head) wave; the proj
    Linear stays with the caller.  Training mode stores q, k, v (packed [n, heads, 3, 49, 32]) and the log-sum-exp rows; the backward
    pass is the unfused one on them (pswin_attn_bwd_ex, then the qkv weight / data gradients): identical kernels and summation order
    to the unfused path."""

    @staticmethod
    def forward(ctx, x, w_qkv, b_qkv, alpha, beta, dist, mask, heads, scale, n_bias_windows, wq_lp, wq_lp_t, grad_mode):
        x = x.contiguous()
        rows, C = x.shape
        n = rows // WTOK
        assert rows % WTOK == 0 and C == heads * _lib.HEAD_DIM
        wq = (wq_lp if wq_lp is not None else w_qkv.to(x.dtype)).contiguous()
        bq = None if b_qkv is None else b_qkv.detach().float().contiguous()
        alpha_c = alpha.detach().contiguous() if dist is not None else None
        beta_c = beta.detach().contiguous()
        save = grad_mode and any(ctx.needs_input_grad)
        y = torch.empty_like(x)
        qkv = lse = None
        if save:
            qkv = torch.empty(n, heads, 3, WTOK, _lib.HEAD_DIM, dtype=x.dtype, device=x.device)
            lse = torch.empty(n, heads, WPAD, dtype=torch.float32, device=x.device)
        flops = n * (2 * WTOK * C * 3 * C + heads * 4 * WTOK * WTOK * _lib.HEAD_DIM)
        call("pswin_qkv_attn_fused_fwd", x, ptr(x), ptr(wq), ptr(bq), ptr(None if dist is None else dist.fwd), 0 if dist is None else dist.n,
             ptr(alpha_c), ptr(beta_c), ptr(None if mask is None else mask.fwd), 0 if mask is None else mask.n, ptr(y), ptr(qkv), ptr(lse), n,
             n_bias_windows, C, heads, float(scale), dtype_code(x), algo_bytes=rows * C * 2 * (5 if save else 2), algo_flops=flops)
        if save:
            ctx.save_for_backward(x, qkv, lse, wq, alpha_c, beta_c, wq_lp_t)
            ctx.params = (w_qkv, b_qkv, alpha if dist is not None else None, beta)
            ctx.cfg = (dist, mask, heads, float(scale), n_bias_windows)
        return y

    @staticmethod
    def backward(ctx, datt):
        x, qkv, lse, wq, alpha_c, beta_c, wq_t = ctx.saved_tensors
        w_qkv, b_qkv, alpha, beta = ctx.params
        dist, mask, heads, scale, nb = ctx.cfg
        C = x.shape[1]
        dqkv, _, _, dalpha, dbeta = attention_backward(qkv, None, None, lse, alpha_c, beta_c, dist, mask, datt, heads, scale, nb, None,
                                                        ctx.needs_input_grad[3] or ctx.needs_input_grad[4], (alpha, beta), packed=True)
        # the K third of d(qkv) sums to zero over every window (rows of dS sum to 0): its bias gradient is not summed
        dx, dwq, dbq = linear_backward(x, wq, dqkv, w_qkv, b_qkv, (C, 2 * C), ctx.needs_input_grad[0], wq_t)
        return dx, dwq, dbq, dalpha, dbeta, None, None, None, None, None, None, None, None


def window_attention_qkv_fused_supported(x2d, heads):
    return (FUSED_QKV_ATTENTION and x2d.is_cuda and x2d.dim() == 2 and x2d.dtype == torch.bfloat16
            and fused_windows_addressable(x2d.shape[0] // WTOK, x2d.shape[1])
            and bool(_lib.load().pswin_qkv_attn_fused_supported(x2d.shape[1], heads, BF16)))


def window_attention_qkv_fused(x2d, attn, dist, mask, n_bias_windows):
    """attention(qkv(x2d)) -- everything of WindowAttention.forward in front of self.proj -- for window rows x2d [n*49, C] and the
    parameter holder `attn`: see _WindowAttentionQkvFused."""
    lq = attn.qkv.__dict__.get("_lowp")
    return _WindowAttentionQkvFused.apply(x2d, attn.qkv.weight, attn.qkv.bias, attn.sphere_position_alpha_table_Te,
                                          attn.sphere_position_beta_table_Te, _as_tiles(dist), _as_tiles(mask), attn.num_heads, attn.scale,
                                          n_bias_windows, lq[0] if lq is not None else None, attn.qkv.__dict__.get("_lowp_t"),
                                          torch.is_grad_enabled())
