"""SimplePanoSwinTransformer on MI355X: the reference backbone's interface over hand-written gfx950 kernels.

Drop-in boundary (reference: mmdet/models/backbones/simple_panoswin_transformer.py = HOT):
  * registry name ``SimplePanoSwinTransformer`` in mmdet's ``BACKBONES`` (HOT:779), same constructor kwargs and
    defaults (HOT:781-801), ``init_weights(pretrained)``, ``forward(x, pano_ratio_v=None) -> tuple of NCHW fp32
    maps`` (HOT:940-979), ``set_pano_mode`` / ``switch_pano_mode`` (HOT:192-208, 880-883), ``train(mode)``;
  * identical ``state_dict`` keys and shapes, so reference checkpoints load with ``strict=True``.

What runs where: every data-layout step of a block (LayerNorm-ed features -> pano/planar shift -> pad -> window
partition, and the way back with crop, DropPath and the residual add) is one indexed row-copy kernel; the
7x7 attention core (q.k^T, great-circle + relative-position bias, mask, softmax, .v) is one MFMA kernel per
direction; PatchMerging's gather and the pitch module's two static bilinear resamplings are row kernels too
(include/pswin.h).  LayerNorm (fused with the gather / the residual accumulation / the NCHW output), bias + GELU, the
bf16 PatchEmbed stem and the Linear layers of the high-resolution stage run in HIP kernels as well (ops.py, stem.py);
the remaining dense projections and every weight gradient are hipBLASLt GEMMs through PyTorch-ROCm, the fp32 parity
configuration of PatchEmbed uses MIOpen convolutions.  uv coordinates never ride as feature channels
(the reference's C+2 layout, HOT:964): they are a function of position, so the great-circle tables are cached
per shape.

Raises, never falls back: CPU tensors, norm_layer other than LayerNorm.  Options the kernels are not specialised for --
window_size != 7, head_dim != 32, drop_rate / attn_drop_rate != 0 (no configuration in the reference tree uses one) -- run their window
blocks as plain torch ops on the GPU (fallback.py) and say so with a warning at construction.
"""
import math
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint as checkpoint

from . import fallback, geometry, ops, stem
from ._lib import HEAD_DIM, WS, WTOK, PswinError
from .registry import BACKBONES


def _relative_position_index(ws):
    """make_relative_position_index (HOT:95-129); kept as a buffer only for state-dict compatibility."""
    t = torch.arange(ws * ws)
    hi, wi = t // ws, t % ws
    return (hi[:, None] - hi[None, :] + ws - 1) * (2 * ws - 1) + (wi[:, None] - wi[None, :] + ws - 1)


def _make_tables(ws, heads):
    """make_table (HOT:132-150).  The reference's two Parameters alias one tensor on CPU and become
    independent after ``.cuda()``; here they are independent parameters with identical initial values."""
    a = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
    nn.init.trunc_normal_(a, std=.02, a=-2.0, b=2.0)
    b = nn.Parameter(a.detach().clone())
    return a, b


def _drop_path_scale(x, p, training):
    """Per-sample DropPath factor of timm's DropPath: floor(keep + U[0,1)) / keep, or None."""
    if p == 0.0 or not training:
        return None
    keep = 1.0 - p
    return torch.floor(keep + torch.rand(x.shape[0], dtype=torch.float32, device=x.device)) / keep


class DoubleModeModule(object):
    """HOT:192-208."""

    def set_pano_mode(self, pano_mode: bool):
        self.pano_mode = pano_mode

    def switch_pano_mode(self):
        self.set_pano_mode(not self.pano_mode)


class Mlp(nn.Module):
    """HOT:44-61."""

    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)

    def forward(self, x, cd):
        return _linear(F.gelu(_linear(x, self.fc1, cd)), self.fc2, cd)

    def forward_nobias2(self, x, cd):
        """fc2(gelu(fc1 x + b1)) WITHOUT fc2's bias (the caller adds it with the residual): bias + GELU in one HIP pass"""
        x2 = x.to(cd).reshape(-1, x.shape[-1])
        if self.fc1.bias is not None and ops.mlp0_fused_supported(x2, self.fc1.weight.shape[0]):
            # stage 0: one autograd node; its backward runs fc2's data gradient and the GELU backward in one pass
            return ops.mlp0_fused(x2, self.fc1, self.fc2).view(*x.shape[:-1], self.fc2.weight.shape[0])
        if self.fc1.bias is not None and ops.fc1_gelu_supported(x2, self.fc1.weight.shape[0]):
            lp = self.fc1.__dict__.get("_lowp")                    # stage 0: fc1 + bias + GELU in one streaming kernel
            h = ops.fc1_gelu(x2, self.fc1.weight, self.fc1.bias, lp[0] if lp is not None else None)
            h = h.view(*x.shape[:-1], h.shape[-1])
        elif self.fc1.bias is not None and ops.mlp_fused_supported(x2, self.fc1.weight.shape[0]):
            # stages 1-3: fc1 (+ bias + GELU in its epilogue) and fc2 as one autograd node on the tiled HIP GEMM
            return ops.mlp_fused(x2, self.fc1, self.fc2).view(*x.shape[:-1], self.fc2.weight.shape[0])
        else:
            y = _linear(x, self.fc1, cd, use_bias=False)
            if cd == torch.bfloat16 and ops.FUSED_GELU_BWD:
                # stages 1-3: bias + GELU + fc2 as one autograd node: its backward runs fc2's data gradient and the GELU
                # backward in one kernel
                return ops.bias_gelu_linear(y, self.fc1.bias, self.fc2)
            h = ops.bias_gelu(y, self.fc1.bias)
        return _linear(h, self.fc2, cd, use_bias=False)


class WindowAttention(nn.Module, DoubleModeModule):
    """Parameter holder of BasicWindowAttention / WindowAttention (HOT:211-323)."""

    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, separate_qkv=False, generic=False):
        super().__init__()
        assert dim % num_heads == 0, "dim must be divisible by num_heads"           # HOT:284
        if not generic and (dim // num_heads != HEAD_DIM or window_size != WS):
            raise PswinError(f"the MI355X attention kernels are specialised for head_dim == {HEAD_DIM} and window_size == {WS}, got "
                             f"dim={dim}, heads={num_heads}, window_size={window_size}")
        self.dim, self.num_heads = dim, num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.register_buffer("relative_position_index_OO", _relative_position_index(window_size))
        self.proj = nn.Linear(dim, dim)
        self.sphere_position_alpha_table_Te, self.sphere_position_beta_table_Te = _make_tables(window_size, num_heads)
        if not separate_qkv:
            self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)


_pick_split, _LinearSplitK, _linear = ops._pick_split, ops._LinearSplitK, ops.linear     # (moved to ops.py; names kept)


class PanoSwinTransformerBlock(nn.Module, DoubleModeModule):
    """HOT:412-536."""

    def __init__(self, dim, num_heads, window_size=7, shift_size=0, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop_path=0., pano_mode=True, drop=0., attn_drop=0.):
        super().__init__()
        assert 0 <= shift_size < window_size, "shift_size must in 0-window_size"
        self.dim, self.num_heads, self.shift_size, self.drop_path_p = dim, num_heads, shift_size, float(drop_path)
        self.window_size, self.drop, self.attn_drop = window_size, float(drop), float(attn_drop)
        # the options the HIP kernels are not specialised for: this block then runs as plain torch ops (fallback.py)
        self.generic = window_size != WS or dim // num_heads != HEAD_DIM or self.drop > 0. or self.attn_drop > 0.
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, window_size, num_heads, qkv_bias, qk_scale, generic=self.generic)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.pano_mode = pano_mode

    def forward(self, x, H, W, cd, dp_scales=None, pre=None, nxt=None, nxt_scales=None, end_norm=None):
        """dp_scales: this block's two DropPath factor vectors ([2, B], from one batched draw for the whole network, see
        SimplePanoSwinTransformer.forward) or None: draw them here.
        pre: (win, x) = this block's norm1 + shift + pad + partition already done by the previous block's closing kernel.
        nxt (+ nxt_scales): the next block of the stage if this block's closing residual add should also run ITS norm1 + partition
        (ops.scatter_add_layer_norm(out=...)); the return value is then that block's `pre` instead of the residual stream.
        end_norm: the stage's output LayerNorm if this block closes the stage and its residual add should run together with that norm
        (ops.scatter_add_layer_norm_nchw); the return value is then (normed NCHW map, residual stream), or (None, residual stream)
        where the fused form does not apply."""
        if pre is None:
            B, S, C = x.shape
        else:
            B, S, C = pre[1].shape
        assert S == H * W, "input feature has wrong size"
        if self.generic:                          # window_size != 7 / head_dim != 32 / dropout: plain torch ops
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(cd == torch.bfloat16 and x.is_cuda)):
                out = fallback.block_forward(self, x, H, W, self.attn_drop, self.drop).float()
            return (None, out) if end_norm is not None else out
        dev = x.device if pre is None else pre[1].device
        pano = bool(self.pano_mode)
        wmap, inv, nW = ops.window_maps(pano, H, W, self.shift_size, dev)
        if pano:
            dist, mask = ops.window_dist_tiles(H, W, self.shift_size, dev), None  # no mask in pano mode (HOT:698-699)
        else:
            dist = None
            mask = ops.planar_mask_tiles(H, W, self.shift_size, dev) if self.shift_size else None   # HOT:474
        a = self.attn
        n1 = self.norm1                                                           # norm1 + shift + pad + partition
        # bf16 path: the proj / fc2 biases ride on the residual row kernels and fc1's on the GELU kernel (no GEMM
        # epilogues, no column-sum passes for their gradients); fp32 (parity) path: plain F.linear with bias
        fuse = cd != torch.float32
        s1, s2 = self.drop_path_scales(x if pre is None else pre[1], dp_scales)
        if pre is not None:
            win, x = pre
        else:
            # the second result is x itself: using it for the shortcut folds the shortcut's gradient into the LN backward kernel
            win, x = ops.layer_norm_gather(x, n1.weight, n1.bias, n1.eps, wmap, inv, cd, passthrough=True,
                                           res_bias=a.proj.bias if fuse else None, res_scale=s1)   # [B, nW*49, C]
        # the K third of d(qkv) sums to zero over every window (rows of dS sum to 0): its bias gradient is not summed
        if fuse and ops.FUSED_WINDOW_ATTENTION and ops.window_attention_fused_supported(win.view(-1, C), a.num_heads):
            # C = 96: qkv Linear, attention and proj Linear of a window in one kernel, weights resident in LDS
            att = ops.window_attention_fused(win.view(-1, C), a, dist, mask, nW).view(B, nW * WTOK, C)
        elif fuse and ops.window_attention_qkv_fused_supported(win.view(-1, C), a.num_heads):
            # C = 192 / 384: qkv Linear + attention core of a (window, head) in one kernel (the head's weight rows resident in LDS)
            att = ops.window_attention_qkv_fused(win.view(-1, C), a, dist, mask, nW)
            att = _linear(att, a.proj, cd, use_bias=False).view(B, nW * WTOK, C)
        else:
            qkv = _linear(win.view(-1, C), a.qkv, cd, zero_bias_cols=(C, 2 * C))      # [B*nW*49, 3C]
            att = ops.window_attention(qkv, a.sphere_position_alpha_table_Te, a.sphere_position_beta_table_Te, dist, mask,
                                       a.num_heads, a.scale, nW)
            att = _linear(att, a.proj, cd, use_bias=not fuse).view(B, nW * WTOK, C)
        n2 = self.norm2
        if x.dtype == torch.float32 and C <= 1024:
            # shortcut + DropPath(attn) and norm2 of the sum in one kernel (the sum is not read back by a LayerNorm pass); its backward
            # kernel also writes the window gather of the shortcut sum's gradient (in_pads: the zero slots of this block's map)
            h, x = ops.scatter_add_layer_norm(att, x, wmap, inv, s1, a.proj.bias if fuse else None, n2.weight, n2.bias,
                                              n2.eps, cd, res_bias=self.mlp.fc2.bias if fuse else None, res_scale=s2,
                                              in_pads=ops.window_pads(pano, H, W, self.shift_size, dev))
        else:
            x = ops.window_scatter_add(att, x, wmap, inv, s1, a.proj.bias if fuse else None, True)
            h, x = ops.layer_norm_gather(x, n2.weight, n2.bias, n2.eps, out_dtype=cd, passthrough=True,
                                         res_bias=self.mlp.fc2.bias if fuse else None, res_scale=s2)
        y = self.mlp.forward_nobias2(h, cd) if fuse else self.mlp(h, cd)
        ident = ops.identity_map(S, dev)
        if nxt is not None:
            # x + DropPath(mlp) AND the next block's norm1 + shift + pad + partition in one kernel: (its windows, the new residual stream)
            npano = bool(nxt.pano_mode)
            _, inv_n, nW_n = ops.window_maps(npano, H, W, nxt.shift_size, dev)
            s1n = nxt.drop_path_scales(x, nxt_scales)[0]
            return ops.scatter_add_layer_norm(y, x, ident, None, s2, self.mlp.fc2.bias, nxt.norm1.weight, nxt.norm1.bias, nxt.norm1.eps, cd,
                                              res_bias=nxt.attn.proj.bias, res_scale=s1n,
                                              out=(inv_n, nW_n * WTOK, ops.window_pads(npano, H, W, nxt.shift_size, dev)))
        if end_norm is not None:
            if fuse and ops.scatter_add_layer_norm_nchw_supported(y, x):
                return ops.scatter_add_layer_norm_nchw(y, x, s2, self.mlp.fc2.bias, end_norm.weight, end_norm.bias, end_norm.eps, H, W)
            return None, ops.window_scatter_add(y, x, ident, ident, s2, self.mlp.fc2.bias if fuse else None, True)
        return ops.window_scatter_add(y, x, ident, ident, s2, self.mlp.fc2.bias if fuse else None, True)   # x + DropPath(mlp): one row kernel

    def drop_path_scales(self, x, dp_scales):
        """(s1, s2): the two per-sample DropPath factor vectors of this block (HOT:533, 536), from the batched draw or drawn here"""
        if dp_scales is not None and self.drop_path_p > 0.0 and self.training:
            return dp_scales[0], dp_scales[1]
        return _drop_path_scale(x, self.drop_path_p, self.training), _drop_path_scale(x, self.drop_path_p, self.training)

    def joins_with(self, nxt, cd, dp_scales_given, C):
        """may this block's closing residual add also run `nxt`'s norm1 + partition?  bf16 path, fp32 residual stream, both blocks on the
        kernels, and the next block's DropPath factors known here (batched draw, or no DropPath to draw)"""
        return (ops.LN_FUSED_MOVES and isinstance(nxt, PanoSwinTransformerBlock) and not self.generic and not nxt.generic and cd != torch.float32
                and C <= 1024 and (dp_scales_given or nxt.drop_path_p == 0.0 or not nxt.training))


class PitchAttentionModule(WindowAttention):
    """HOT:990-1237; appended to a stage whose depth is odd (HOT:636-647)."""

    def __init__(self, dim, num_heads, window_size=7, qkv_bias=True, qk_scale=None, mlp_ratio=4., np_v=-0.0001,
                 pano_mode=True):
        super().__init__(dim, window_size, num_heads, qkv_bias, qk_scale, separate_qkv=True)
        self.window_size = window_size
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.norm2 = nn.LayerNorm(dim)
        self.norm1 = nn.LayerNorm(dim)
        self.q_linear = nn.Linear(dim, dim, bias=qkv_bias)
        self.k_linear = nn.Linear(dim, dim, bias=qkv_bias)
        self.v_linear = nn.Linear(dim, dim, bias=qkv_bias)
        self.register_buffer("np_uv", torch.Tensor([1.0, np_v]) * math.pi)
        self.pano_mode = pano_mode
        self._static = {}

    def _tables(self, H, W, dev):
        """Static resampling tables, the rotated-window uv and the q-vs-rotated-k distance table (per shape)."""
        key = (H, W, str(dev))
        if key not in self._static:
            t = geometry.pitch_tables(H, W, self.window_size, self.np_uv)
            t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in t.items()}
            wmap, _, nW = ops.window_maps(False, H, W, 0, dev)
            uv = ops.uv_grid(H, W, dev)
            uv4 = torch.cat([uv, torch.zeros_like(uv)], -1)[None].contiguous()           # rows of 4 for the row kernel
            uv_rot = ops.interp_rows(ops.interp_rows(uv4, t["idx1"], t["w1"]), t["idx2"], t["w2"])[0, :, :2]
            uv_win = ops.gather_uv(uv, wmap)
            t["dist"] = ops.Tiles(ops.haversine_windows(uv_win.view(nW, WTOK, 2), uv_rot.contiguous().view(nW, WTOK, 2)))
            self._static[key] = t
        return self._static[key]

    def forward(self, x, H, W, cd):
        B, S, C = x.shape
        assert S == H * W, "input feature has wrong size"
        dev = x.device
        pano = bool(self.pano_mode)
        wmap, inv, nW = ops.window_maps(False, H, W, 0, dev)
        xn = ops.layer_norm_gather(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        win = ops.window_gather(xn, wmap, inv, cd).view(-1, C)
        if pano:
            t = self._tables(H, W, dev)
            rot = ops.interp_rows(ops.interp_rows(xn, t["idx1"], t["w1"]), t["idx2"], t["w2"])   # [B, nW*49, C]
            win_rot, dist = rot.view(-1, C), t["dist"]
        else:
            win_rot, dist = win, None
        q, k, v = _linear(win, self.q_linear, cd), _linear(win_rot, self.k_linear, cd), _linear(win, self.v_linear, cd)
        att = ops.window_attention(q, self.sphere_position_alpha_table_Te, self.sphere_position_beta_table_Te, dist,
                                   None, self.num_heads, self.scale, nW, k=k, v=v)
        att = _linear(att, self.proj, cd).view(B, nW * WTOK, C)
        # the reference overwrites its own shortcut with LN(x) (in-place norm on a view, HOT:1154-1155): residual = xn
        x = ops.window_scatter_add(att, xn, wmap, inv, None)
        h, x = ops.layer_norm_gather(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, out_dtype=cd, passthrough=True)
        y = self.mlp(h, cd)
        ident = ops.identity_map(S, dev)
        return ops.window_scatter_add(y, x, ident, ident, None)


class PatchMerging(nn.Module):
    """HOT:539-576."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def forward(self, x, H, W, cd):
        B, S, C = x.shape
        assert S == H * W, "input feature has wrong size"
        g = ops.layer_norm_patch_merge(x, self.norm.weight, self.norm.bias, self.norm.eps, H, W, cd)
        # the fp32 residual stream of the next stage straight from the GEMM's epilogue (pswin_gemm_nt_f32) where the tiled kernel runs
        return _linear(g, self.reduction, cd, out_f32=(x.dtype == torch.float32)).to(x.dtype)


class BasicLayer(nn.Module, DoubleModeModule):
    """HOT:579-724."""

    def __init__(self, dim, depth, num_heads, window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop_path=0., downsample=None, use_checkpoint=False, pano_mode=True, drop=0., attn_drop=0.):
        super().__init__()
        self.use_checkpoint = use_checkpoint
        blocks = [PanoSwinTransformerBlock(dim, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2,
                                           mlp_ratio, qkv_bias, qk_scale,
                                           drop_path[i] if isinstance(drop_path, list) else drop_path, pano_mode, drop, attn_drop)
                  for i in range(depth - depth % 2)]
        if depth % 2 and blocks and blocks[0].generic:
            raise PswinError("an odd stage depth appends a PitchAttentionModule (HOT:636-647), which has no generic form: window_size == 7, "
                             "head_dim == 32 and dropout 0 are required for it")
        if depth % 2:
            blocks.append(PitchAttentionModule(dim, num_heads, window_size, qkv_bias, qk_scale, mlp_ratio,
                                               pano_mode=pano_mode))
        self.blocks = nn.ModuleList(blocks)
        self.downsample = downsample(dim) if downsample is not None else None
        self.pano_mode = pano_mode

    def set_pano_mode(self, pano_mode=True):
        self.pano_mode = pano_mode
        for blk in self.blocks:
            blk.set_pano_mode(pano_mode)

    def forward(self, x, H, W, cd, out_norm=None, dp_scales=None):
        """-> (stage output [normed by out_norm if given], H, W, input of the next stage, its H, W)"""
        pre = y_end = None
        for i, blk in enumerate(self.blocks):
            is_blk = isinstance(blk, PanoSwinTransformerBlock)
            sc = dp_scales[i] if (dp_scales is not None and is_blk) else None
            if self.use_checkpoint or not is_blk:
                x = checkpoint.checkpoint(blk, x, H, W, cd, *(() if sc is None else (sc,)), use_reentrant=False) if self.use_checkpoint \
                    else blk(x, H, W, cd)
                continue
            nxt = self.blocks[i + 1] if i + 1 < len(self.blocks) else None
            if nxt is not None and not blk.joins_with(nxt, cd, dp_scales is not None, x.shape[-1] if pre is None else pre[1].shape[-1]):
                nxt = None
            last = nxt is None and i + 1 == len(self.blocks) and out_norm is not None
            out = blk(x, H, W, cd, sc, pre, nxt, dp_scales[i + 1] if (nxt is not None and dp_scales is not None) else None,
                      out_norm if last else None)
            if nxt is not None:
                pre, x = out, None
            elif last:
                pre, (y_end, x) = None, out         # the stage's closing residual add ran together with its output norm (or y_end is None)
            else:
                pre, x = None, out
        y = x
        if out_norm is not None and y_end is not None:
            y = y_end
        elif out_norm is not None:                  # output norm first: the downsample branch's gradient then joins
            if self.downsample is not None:         # the stream inside the norm's backward kernel; NCHW written directly
                y, x = ops.layer_norm_nchw(x, out_norm.weight, out_norm.bias, out_norm.eps, H, W, passthrough=True)
            else:
                y = ops.layer_norm_nchw(x, out_norm.weight, out_norm.bias, out_norm.eps, H, W)
        if self.downsample is None:
            return y, H, W, x, H, W
        return y, H, W, self.downsample(x, H, W, cd), (H + 1) // 2, (W + 1) // 2


class _ApeAdd(torch.autograd.Function):
    """x + Linear(feat)[None]  (the absolute position encoding, HOT:926-934: feat [S, 5] is input independent).
    Backward without a framework two-pass reduction: the bias gradient (a sum over all S tokens) goes through
    pswin_colsum.  torch's global reductions return stale results from the second replay of a captured hipGraph on this
    stack (the semaphore memset node is not re-run: tools/repro_graph_stale_reduction.py), which silently corrupted this one gradient."""

    @staticmethod
    def forward(ctx, x, feat, weight, bias):
        ctx.save_for_backward(feat, weight)
        ctx.bias = bias
        return x + F.linear(feat, weight, bias)[None]

    @staticmethod
    def backward(ctx, g):
        feat, weight = ctx.saved_tensors
        gs = g.sum(0) if g.shape[0] > 1 else g[0]              # [S, C]: a per-element sum over the batch (one pass)
        dw = gs.t() @ feat                                      # [C, 5]
        db = ops.colsum(gs.contiguous(), owners=(ctx.bias,))
        return g, None, dw, db


class _ApeRows(torch.autograd.Function):
    """Linear(feat) [S, C] alone: the rows PatchEmbed's LayerNorm kernel adds while it writes the residual stream
    (ops.layer_norm_gather(add_rows=...)) -- the same arithmetic as _ApeAdd without the pass over [B, S, C]; the gradient it
    receives is already summed over the batch."""

    @staticmethod
    def forward(ctx, feat, weight, bias):
        ctx.save_for_backward(feat)
        ctx.bias = bias
        return F.linear(feat, weight, bias)

    @staticmethod
    def backward(ctx, gs):
        feat, = ctx.saved_tensors
        gs = gs.contiguous()
        return None, gs.t() @ feat, ops.colsum(gs, owners=(ctx.bias,))


class _ChannelBias(torch.autograd.Function):
    """y + bias over the channel dim of a channels-last NCHW tensor; the bias gradient is a column sum over the
    [N*H*W, C] row view (pswin_colsum).  Keeps the convolution bias out of MIOpen's ConvolutionBackwardBias, whose
    result is garbage (1e34) from the second replay on when the step is captured in a hipGraph (ROCm 7.2)."""

    @staticmethod
    def forward(ctx, y, bias):
        return y + bias.to(y.dtype).view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, dy):
        return dy, ops.colsum_channels(dy)                     # any channel count (narrow rows are zero-padded to 8 columns)


class PatchEmbed(nn.Module):
    """HOT:727-773.  bf16 compute on the default geometry (3 -> 32 -> 64 -> 96, patch 4): the fused HIP stem
    (stem.py / csrc/pswin_stem.hip).  Otherwise (fp32 parity configuration, other widths, an image that needs a
    gradient): MIOpen convolutions, channels-last so that the token layout needs no transpose, + the HIP BN/ReLU kernels."""

    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, norm=True):
        super().__init__()
        self.patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.embed_dim = embed_dim
        c = embed_dim // 3
        self.proj = nn.Sequential(
            nn.Conv2d(in_chans, c, kernel_size=3, stride=1, padding=1), nn.BatchNorm2d(c), nn.ReLU(inplace=True),
            nn.Conv2d(c, c * 2, kernel_size=3, stride=1, padding=1), nn.BatchNorm2d(c * 2), nn.ReLU(inplace=True),
            nn.Conv2d(c * 2, embed_dim, kernel_size=self.patch_size, stride=self.patch_size))
        self.norm = nn.LayerNorm(embed_dim) if norm else None

    def forward(self, x, cd=torch.float32, add_rows=None):
        """add_rows: a callable (Wh, Ww) -> f32 [Wh * Ww, C] rows added to the normalised tokens inside the LayerNorm kernel (the
        absolute position encoding); it is used -- and the third return value True -- only where that kernel runs."""
        _, _, H, W = x.shape
        ph, pw = self.patch_size
        if W % pw:
            x = F.pad(x, (0, pw - W % pw))
        if H % ph:
            x = F.pad(x, (0, 0, 0, ph - H % ph))
        if stem.stem_supported(self.proj, x, cd):            # fused HIP stem (csrc/pswin_stem.hip): y2 is the only
            B, Wh, Ww = x.shape[0], x.shape[2] // ph, x.shape[3] // pw    # full-resolution tensor ever stored
            tok = stem.stem_forward(self.proj, x, self.training).view(B, Wh * Ww, self.embed_dim)
            if self.norm is not None:
                rows = add_rows(Wh, Ww) if add_rows is not None else None
                tok = ops.layer_norm_gather(tok, self.norm.weight, self.norm.bias, self.norm.eps, out_dtype=torch.float32, add_rows=rows)
                return (tok.float(), Wh, Ww, True) if add_rows is not None else (tok.float(), Wh, Ww)
            return (tok.float(), Wh, Ww, False) if add_rows is not None else (tok.float(), Wh, Ww)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(cd == torch.bfloat16)):
            x = x.contiguous(memory_format=torch.channels_last)                 # bf16: MIOpen NHWC bf16 convolutions
            mods = list(self.proj)
            pending_bias = None
            for i, mod in enumerate(mods):
                if isinstance(mod, nn.Conv2d):
                    x = F.conv2d(x, mod.weight, None, mod.stride, mod.padding)
                    nxt = mods[i + 1] if i + 1 < len(mods) else None
                    if isinstance(nxt, nn.BatchNorm2d) and nxt.num_features % 8 == 0:
                        pending_bias = mod.bias      # cancels inside the BatchNorm: folded into its running mean only
                    else:
                        x = _ChannelBias.apply(x, mod.bias)
                elif isinstance(mod, nn.BatchNorm2d):
                    if mod.num_features % 8 == 0:
                        x = ops.batch_norm_relu(x, mod, self.training, pending_bias)   # BN + ReLU: one HIP op
                        pending_bias = None
                    else:                                                   # odd widths (embed_dim % 24 != 0): MIOpen
                        x = F.relu(mod(x))
                elif not isinstance(mod, nn.ReLU):
                    x = mod(x)
        B, C, Wh, Ww = x.shape
        tok = x.permute(0, 2, 3, 1).reshape(B, Wh * Ww, C)          # free for a channels-last tensor
        if self.norm is not None:                                    # reads bf16 or fp32, writes the fp32 residual stream
            rows = add_rows(Wh, Ww) if add_rows is not None else None
            tok = ops.layer_norm_gather(tok, self.norm.weight, self.norm.bias, self.norm.eps, out_dtype=torch.float32, add_rows=rows)
            return (tok.float(), Wh, Ww, True) if add_rows is not None else (tok.float(), Wh, Ww)
        return (tok.float(), Wh, Ww, False) if add_rows is not None else (tok.float(), Wh, Ww)


@BACKBONES.register_module()
class SimplePanoSwinTransformer(nn.Module, DoubleModeModule):
    """PanoSwin backbone (HOT:779-983) with the reference's constructor, plus one keyword:

    compute_dtype: torch.float32 (default: fp32 end to end, the parity configuration) or torch.bfloat16 (bf16
    GEMM / attention operands with fp32 accumulation, softmax, LayerNorm statistics and residual stream - the
    analogue of the reference's apex O1 setting, mmdet/apis/train.py:82-88).
    """

    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, depths=[2, 2, 7, 2], num_heads=[3, 6, 12, 24],
                 window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0.2, norm_layer=nn.LayerNorm, ape=False, patch_norm=True, out_indices=(0, 1, 2, 3),
                 frozen_stages=-1, use_checkpoint=False, pano_mode=True, compute_dtype=torch.float32):
        super().__init__()
        self.drop_rate = float(drop_rate)
        generic = [f"window_size={window_size}"] if window_size != WS else []
        generic += [f"head_dim={embed_dim // num_heads[0]}"] if embed_dim // num_heads[0] != HEAD_DIM else []
        generic += [f"drop_rate={drop_rate}"] if drop_rate else []
        generic += [f"attn_drop_rate={attn_drop_rate}"] if attn_drop_rate else []
        if generic:
            warnings.warn("SimplePanoSwinTransformer: " + ", ".join(generic) + " is outside what the MI355X kernels are specialised for "
                          "(window 7, head_dim 32, dropout 0: every configuration of the reference tree); the window blocks of this model "
                          "run as plain torch ops on the GPU (fallback.py), not on the hand-written kernels")
        if norm_layer is not nn.LayerNorm and norm_layer != "LN" and norm_layer is not None:
            raise PswinError("norm_layer must be nn.LayerNorm")
        if isinstance(compute_dtype, str):
            compute_dtype = {"float32": torch.float32, "fp32": torch.float32, "bfloat16": torch.bfloat16,
                             "bf16": torch.bfloat16}[compute_dtype]
        if compute_dtype not in (torch.float32, torch.bfloat16):
            raise PswinError("compute_dtype must be float32 or bfloat16")
        self.compute_dtype = compute_dtype
        self.num_layers, self.embed_dim, self.ape, self.patch_norm = len(depths), embed_dim, ape, patch_norm
        self.out_indices, self.frozen_stages = out_indices, frozen_stages      # frozen_stages is ignored, as in HOT:833
        self.patch_embed = PatchEmbed(patch_size, in_chans, embed_dim, patch_norm)
        if self.ape:
            self.abs_encoder = nn.Linear(5, embed_dim)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(BasicLayer(int(embed_dim * 2 ** i), depths[i], num_heads[i], window_size, mlp_ratio,
                                          qkv_bias, qk_scale, dpr[sum(depths[:i]):sum(depths[:i + 1])],
                                          PatchMerging if i < self.num_layers - 1 else None, use_checkpoint,
                                          pano_mode, drop_rate, attn_drop_rate))
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        for i in out_indices:
            self.add_module(f"norm{i}", nn.LayerNorm(self.num_features[i]))
        self.set_pano_mode(pano_mode)

    def set_pano_mode(self, pano_mode=True):
        self.pano_mode = pano_mode
        for layer in self.layers:
            layer.set_pano_mode(pano_mode)

    LATE_STAGES = 2          # the stages from this index on form the first gradient group (dp.py: reduced under the rest)

    def grad_groups(self):
        """Trainable parameters in the order a backward pass finishes their gradients, as two groups for
        dp.GradReducer / dp.split_parameters: [norm3, layers.3, norm2, layers.2] (90 % of the bytes; their all-reduce runs
        under the backward pass of the second group) and [norm1, layers.1, norm0, layers.0, abs_encoder, patch_embed]."""
        def stage(i):
            ps = list(getattr(self, f"norm{i}").parameters()) if i in self.out_indices else []
            return ps + list(reversed(list(self.layers[i].parameters())))
        cut = min(self.LATE_STAGES, self.num_layers)
        late = [p for i in range(self.num_layers - 1, cut - 1, -1) for p in stage(i)]
        early = [p for i in range(cut - 1, -1, -1) for p in stage(i)]
        if self.ape:
            early += list(reversed(list(self.abs_encoder.parameters())))
        early += list(reversed(list(self.patch_embed.parameters())))
        return [[p for p in late if p.requires_grad], [p for p in early if p.requires_grad]]

    def init_weights(self, pretrained=None):
        """HOT:885-907."""
        def _init_weights(m):
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02, a=-2.0, b=2.0)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)

        if isinstance(pretrained, str):
            self.apply(_init_weights)
            from .checkpoint import load_checkpoint
            load_checkpoint(self, pretrained, strict=False)
        elif pretrained is None:
            self.apply(_init_weights)
        else:
            raise TypeError('pretrained must be a str or None')

    @torch.no_grad()
    def _refresh_lowp(self, cd, for_backward=False):
        """bf16 copies of every Linear weight / bias on the path, refreshed by one multi-tensor cast per forward
        (instead of ~100 small cast kernels); plain attributes, never part of the state dict."""
        fp = self.__dict__.get("_flat_pair")
        if fp is not None and fp[1].dtype == cd:
            # optim.FlatAdamW writes the low-precision copy with the update itself; the copy is then skipped here -- but only while
            # nothing else has touched the weights since (load_state_dict, init_weights, another optimizer, ...): see
            # _weights_signature.  The optimizer is held weakly: once it is gone every forward pass copies again.
            ext = self.__dict__.get("_lowp_external")
            if ext is not None and ext() is None:
                ext = self.__dict__["_lowp_external"] = None
            if ext is None or self.__dict__.get("_lowp_sig") != self._weights_signature():
                fp[1].copy_(fp[0])                  # every master weight -> its low-precision view, one kernel
                self.__dict__["_lowp_sig"] = self._weights_signature()
            if for_backward:
                self._refresh_transposed(cd)
            return
        lins = [m for m in self.modules() if isinstance(m, nn.Linear) and m is not getattr(self, "abs_encoder", None)]
        src, dst = [], []
        for m in lins:
            lp = m.__dict__.get("_lowp")
            if lp is None or lp[0].device != m.weight.device or lp[0].dtype != cd:
                lp = (torch.empty_like(m.weight, dtype=cd), None if m.bias is None else torch.empty_like(m.bias, dtype=cd))
                m.__dict__["_lowp"] = lp
            src.append(m.weight)
            dst.append(lp[0])
            if m.bias is not None:
                src.append(m.bias)
                dst.append(lp[1])
        torch._foreach_copy_(dst, src)
        if for_backward:
            self._refresh_transposed(cd)

    def _weights_signature(self):
        """Changes when a master weight is written through PyTorch: the version counters of the flat buffer (an optimizer on the
        flat parameter) and of every parameter (load_state_dict, nn.init, ``p.copy_``).  optim.FlatAdamW's kernel writes through raw
        pointers and bumps neither -- which is the point: after its step the shadow is already fresh.  Writes through ``p.data`` are
        invisible to autograd's counters; such callers (dp.GradReducer.broadcast_parameters does) call mark_weights_changed()."""
        fp = self.__dict__["_flat_pair"]
        return (fp[0]._version, sum(p._version for p in self.parameters()), self.__dict__.get("_lowp_epoch", 0))

    def mark_weights_changed(self):
        """Tell the model that its weights were modified behind autograd's back (``p.data`` writes, raw pointers): the next forward
        pass refreshes the low-precision shadows whatever optimizer is attached."""
        self.__dict__["_lowp_epoch"] = self.__dict__.get("_lowp_epoch", 0) + 1

    @torch.no_grad()
    def _refresh_transposed(self, cd):
        """[K, N] copies of this step's bf16 Linear weights for the layers whose data gradient runs on pswin_gemm_nt (the
        kernel wants both operands contraction-contiguous): one batched transpose launch per training step."""
        if cd != torch.bfloat16 or not ops.GEMM_NT:
            return
        pairs = self.__dict__.get("_lowp_t_pairs")
        if pairs is None or any(lp.data_ptr() != m.__dict__["_lowp"][0].data_ptr() for m, (lp, _) in pairs):
            pairs = []
            for m in self.modules():
                lp = m.__dict__.get("_lowp") if isinstance(m, nn.Linear) else None
                if lp is None:
                    continue
                N, K = lp[0].shape
                if N % 64 == 0 and K % 192 == 0 and ops._lib.load().pswin_gemm_nt_supported(8192, N, K):
                    t = torch.empty(K, N, dtype=cd, device=lp[0].device)
                    m.__dict__["_lowp_t"] = t
                    pairs.append((m, (lp[0], t)))
            self.__dict__["_lowp_t_pairs"] = pairs
        ops.transpose_weights([p for _, p in pairs])

    def _attach_flat_lowp(self, flat_master, flat_lowp):
        """Called by dp.GradReducer.flatten_parameters: all parameters are views of `flat_master`; make the low-precision
        shadows of the Linear layers views of `flat_lowp` at the same offsets, so that one cast refreshes them all."""
        base = flat_master.data_ptr()
        for m in self.modules():
            if isinstance(m, nn.Linear) and m is not getattr(self, "abs_encoder", None):
                views = []
                for p in (m.weight, m.bias):
                    if p is None:
                        views.append(None)
                        continue
                    off = (p.data_ptr() - base) // 4
                    assert 0 <= off and off + p.numel() <= flat_master.numel(), "parameter is not a view of the flat buffer"
                    views.append(flat_lowp[off:off + p.numel()].view_as(p))
                m.__dict__["_lowp"] = tuple(views)
        self.__dict__["_flat_pair"] = (flat_master, flat_lowp)

    def forward(self, x_bchw, pano_ratio_v=None):
        if pano_ratio_v is not None:
            warnings.warn("Parameter pano_ratio_v for is deprecated! Please set it to None!")
        if self.pano_mode and x_bchw.shape[3] != x_bchw.shape[2] * 2:
            warnings.warn("PanoSwin is configured in Pano mode, expecting channel3 == 2 * channel2, but get {} and {}, "
                          "probably cause an error".format(x_bchw.shape[3], x_bchw.shape[2]))
        if not x_bchw.is_cuda:
            raise PswinError("SimplePanoSwinTransformer (MI355X build) needs its input on a HIP device")
        cd = self.compute_dtype
        if cd != torch.float32:
            self._refresh_lowp(cd, for_backward=torch.is_grad_enabled())
        if self.pano_mode and self.ape:                                                  # HOT:926-934
            dev = x_bchw.device
            x, Wh, Ww, added = self.patch_embed(x_bchw.float(), cd, add_rows=lambda h, w: _ApeRows.apply(
                ops.abs_pos_features(h, w, dev), self.abs_encoder.weight, self.abs_encoder.bias))
            if not added:                                                                # no PatchEmbed.norm to carry the addition
                x = _ApeAdd.apply(x, ops.abs_pos_features(Wh, Ww, dev), self.abs_encoder.weight, self.abs_encoder.bias)
        else:
            x, Wh, Ww = self.patch_embed(x_bchw.float(), cd)
        if self.drop_rate > 0.:
            x = F.dropout(x, self.drop_rate, self.training)                              # pos_drop (HOT:967), features only
        outs = []
        dp_all = self._draw_drop_path(x) if self.training else None
        for i, layer in enumerate(self.layers):
            nl = getattr(self, f"norm{i}") if i in self.out_indices else None
            y, H, W, x, Wh, Ww = layer(x, Wh, Ww, cd, nl, None if dp_all is None else dp_all[i])
            if nl is not None:
                outs.append(y)                                                          # already [B, C, H, W], HOT:975-977
        return tuple(outs)

    def _draw_drop_path(self, x):
        """All DropPath factors of one forward pass from ONE uniform draw (two per block: HOT:533, 536): per stage a
        [blocks, 2, B] tensor of floor(keep + U) / keep.  The reference draws per call; drawing up front gives the same
        distribution with 4 small kernels per step instead of 3 per branch."""
        keeps = []
        for layer in self.layers:
            for blk in layer.blocks:
                p = getattr(blk, "drop_path_p", 0.0)
                keeps += [1.0 - p, 1.0 - p]
        if not any(k < 1.0 for k in keeps):
            return None
        key = (x.device, tuple(keeps))
        if getattr(self, "_dp_keep_key", None) != key:
            self._dp_keep = torch.tensor(keeps, dtype=torch.float32, device=x.device)[:, None]
            self._dp_keep_key = key
        u = torch.rand(len(keeps), x.shape[0], dtype=torch.float32, device=x.device)
        scales = torch.floor(self._dp_keep + u) / self._dp_keep
        out, at = [], 0
        for layer in self.layers:
            n = len(layer.blocks)
            out.append(scales[at:at + 2 * n].view(n, 2, -1))
            at += 2 * n
        return out

    def train(self, mode=True):
        """The reference's train() forgets to return self (HOT:981-983); nn.Module semantics are kept here."""
        super().train(mode)
        return self
