/*
 * pswin.h -- C ABI of libpswin_hip.so: the MI355X (gfx950) kernels behind the PanoSwin windowed-attention
 * hot path.
 *
 * The reference (1069066484/PanoSwinTransformerObjectDetection) has no native code and no FFI: every
 * step of the path is eager PyTorch inside mmdet/models/backbones/simple_panoswin_transformer.py ("HOT").
 * Each entry point below therefore names the reference *Python* function(s) it replaces (file:line
 * relative to the reference root); INTEGRATION.md shows the ctypes binding a reference maintainer adds.
 *
 * Conventions
 *   - plain C: raw device pointers, explicit sizes, no torch types.  `stream` is a hipStream_t passed as
 *     void* (NULL = the default stream).  The caller owns every buffer including workspaces; the library
 *     allocates nothing, keeps no global state and is thread-safe (the caller sets the device).
 *   - every function returns 0 on success, a negative PSWIN_ERR_* for a rejected argument, or a positive
 *     hipError_t from the launch.  Nothing aborts or throws.
 *   - launches are asynchronous on `stream`; no function synchronises.
 *   - "rows" are feature vectors of C contiguous elements; tensors are dense row-major.
 *   - dtype codes: PSWIN_F32 = 0, PSWIN_BF16 = 1 (bf16 = upper 16 bits of an IEEE f32, RNE conversion).
 *   - window size is 7 (49 tokens) and head_dim is 32 in every attention entry point, as in every
 *     configuration the reference ships (the swin model configs under configs/_base_/models: window_size=7, C/heads=32).
 */
#ifndef PSWIN_H_
#define PSWIN_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSWIN_ABI_VERSION 2

#define PSWIN_F32 0
#define PSWIN_BF16 1

#define PSWIN_MODE_PLANAR 0
#define PSWIN_MODE_PANO 1

#define PSWIN_OK 0
#define PSWIN_ERR_ARG (-1)         /* null pointer, non-positive size, misaligned or inconsistent sizes */
#define PSWIN_ERR_UNSUPPORTED (-2) /* valid request outside what the kernels are specialised for */

#define PSWIN_WS 7        /* window size */
#define PSWIN_WTOK 49     /* tokens per window */
#define PSWIN_WPAD 64     /* tokens per window padded to the MFMA tile */
#define PSWIN_HEAD_DIM 32

int pswin_version(void);

/* ------------------------------------------------------------------------------------------------
 * Index maps (bit-exact integer work)
 * ---------------------------------------------------------------------------------------------- */

/* Host helper: padded grid of the window layout.
 * pano  : the north-south layout [2H, ceil(W/2)] padded to multiples of 7 (HOT:337-353, 486-491)
 * planar: [H, W] padded to multiples of 7 (HOT:486-491).
 * Writes Hp, Wp and the number of windows per image.  Pure host arithmetic, no launch. */
int pswin_window_grid(int mode, int H, int W, int* Hp, int* Wp, int* n_windows);

/* Window map and its inverse, on the device.
 * Replaces the chain WindowTransition.forward -> pad_x -> window_partition (HOT:376-409, 486-491, 64-75)
 * and, for `inv`, crop -> WindowTransition.forward(reverse=True) after window_reverse (HOT:78-92, 516-528).
 *   map[slot] , slot = window*49 + token : flat source token h*W+w, or -1 for a zero (padding) slot
 *   inv[h*W+w]                            : the slot that holds this token (every token has exactly one)
 * pano  : roll W by +shift, east-west -> north-south fold, roll H by +shift   (closed form)
 * planar: pad, then roll by (-shift, -shift)                                  (closed form)
 * map: int32 [n_windows*49]; inv: int32 [H*W]. */
int pswin_window_map(int mode, int H, int W, int shift, int32_t* map, int32_t* inv, void* stream);

/* Planar shifted-window attention mask, BasicLayer._get_attention_mask (HOT:664-688):
 * mask[w][i][j] = 0 if tokens i, j of window w lie in the same of the 9 shift regions else -100.0.
 * mask: f32 [n_windows, 49, 49] for the planar grid of (H, W). */
int pswin_planar_mask(int H, int W, int shift, float* mask, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Geometry (fp32)
 * ---------------------------------------------------------------------------------------------- */

/* make_uv_hw2 (HOT:153-189): uv[(y*W+x)*2 + {0,1}] = (u, v), u = x*gap - pi + gap/2, v = y*gap - pi/2 + gap/2,
 * gap = pi/H, evaluated in fp32 in exactly that order (no FMA contraction) -> bit-identical to the reference. */
int pswin_uv_grid(int H, int W, float* uv, void* stream);

/* SimplePanoSwinTransformer._pano_abs_position input features (HOT:926-932):
 * feat[t*5 + 0..4] = (sin u sin v, cos u sin v, cos v, u, v) for n tokens. */
int pswin_abs_pos_features(const float* uv, int n, float* feat, void* stream);

/* uv of every window slot: uv_win[slot] = uv[map[slot]] or (0, 0) in padding slots (the reference carries
 * uv as two feature channels, so zero padding zeroes them: HOT:504-513). */
int pswin_gather_uv(const float* uv, const int32_t* map, int n_slots, float* uv_win, void* stream);

/* haversine22 (lzx/models/great_circle.py:71-86) per window:
 * dist[w][i][j] = 2 asin(sqrt(sin^2(|v2_j - v1_i|/2) + cos v2_j cos v1_i sin^2((u2_j - u1_i)/2))).
 * uv1, uv2: f32 [n_windows, 49, 2]; dist: f32 [n_windows, 49, 49]. */
int pswin_haversine_windows(const float* uv1, const float* uv2, int n_windows, float* dist, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Row movers (HBM-bound gathers; the copies of window_partition / window_reverse / roll / flip / cat /
 * pad that the reference materialises one by one: SURVEY appendix B)
 * ---------------------------------------------------------------------------------------------- */

/* win[b][slot][:] = map[slot] >= 0 ? (scale ? scale[b] : 1) * x[b][map[slot]][:] : 0
 * Forward of window partition (x = LayerNorm-ed features) and, with map = the forward map and scale = the
 * per-sample DropPath factor, backward of pswin_window_scatter_add.
 * x: [B, S, C] of x_dtype; win: [B, n_slots, C] of win_dtype; C % 8 == 0. */
int pswin_window_gather(const void* x, int x_dtype, const int32_t* map, const float* scale, void* win,
                        int win_dtype, int B, int S, int n_slots, int C, void* stream);

/* out[b][t][:] = (resid ? resid[b][t][:] : 0) + (scale ? scale[b] : 1) * (win[b][inv[t]][:] + (bias ? bias[:] : 0))
 * window_reverse + crop + reverse transition + residual add + DropPath scaling (HOT:483, 516-533), and,
 * with resid = NULL, backward of pswin_window_gather.  bias (f32 [C], may be NULL): the bias of the Linear that
 * produced win (proj, HOT:309; fc2, HOT:58), added here so that the GEMM runs without an epilogue and the bias
 * gradient comes out of pswin_ln_gather_bwd (dres_sum) instead of a column-sum pass over the GEMM's output gradient.
 * win: [B, n_slots, C] win_dtype; resid, out: [B, S, C] of x_dtype. */
int pswin_window_scatter_add(const void* win, int win_dtype, const int32_t* inv, const void* resid,
                             const float* scale, const float* bias, void* out, int x_dtype, int B, int S, int n_slots,
                             int C, void* stream);

/* LayerNorm fused into the window gather: y[b][slot][:] = LN(x[b][map[slot]][:]) * gamma + beta, zero rows in the
 * padding slots (the reference pads AFTER norm1: HOT:504, 512).  Replaces norm1 + WindowTransition + pad_x +
 * window_partition (HOT:503-513).  With map == NULL (then n_out must equal S) it is a plain row LayerNorm: norm2
 * (HOT:534), PatchEmbed.norm (HOT:771), the output norms (HOT:975-976).
 * Statistics (biased variance, eps inside the sqrt, as nn.LayerNorm) are fp32 and stored per SOURCE token:
 * mean, rstd: f32 [B, S].  x: [B, S, C] x_dtype; y: [B, n_out, C] y_dtype; gamma, beta: f32 [C]; 8 <= C <= 2048, C % 8 == 0. */
int pswin_ln_gather_fwd(const void* x, int x_dtype, const int32_t* map, const float* gamma, const float* beta, float eps,
                        void* y, int y_dtype, float* mean, float* rstd, int B, int S, int n_out, int C, void* stream);

/* The same with one more f32 row added to every normalised row: y[b][t][:] = LN(x[b][t][:]) * gamma + beta + add_rows[t][:].
 * add_rows: f32 [S, C] (NULL: nothing added) -- the absolute position encoding of HOT:926-934 (abs_encoder(xyzuv), the same for
 * every image) added to the PatchEmbed.norm output (HOT:771, 934) in the pass that writes it; map must be NULL with add_rows. */
int pswin_ln_gather_fwd_add(const void* x, int x_dtype, const int32_t* map, const float* gamma, const float* beta, float eps,
                            const float* add_rows, void* y, int y_dtype, float* mean, float* rstd, int B, int S, int n_out, int C,
                            void* stream);

/* Its backward: for every token (b, t), dy row = dy[b][inv ? inv[t] : t] and
 *   dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma, xhat = (x - mean) * rstd;
 *   dgamma = sum dy * xhat, dbeta = sum dy (block partials in `workspace`, reduced in a fixed order).
 * dres (may be NULL; f32 [B, S, C], needs x_dtype f32): the gradient reaching x along the residual shortcut
 * (x + DropPath(f(norm(x))), HOT:533-536); it is added into dx here, which replaces autograd's separate accumulation pass.
 * dres_sum (may be NULL; f32 [C]) = sum_{b,t} res_scale[b] * dres[b][t][:] (res_scale: f32 [B] DropPath factors or NULL = 1):
 * the gradient of a bias that pswin_window_scatter_add added on the branch whose shortcut dres came along (the proj /
 * fc2 bias of the block, HOT:309, 58), obtained from the pass that reads dres anyway.
 * dy: [B, n_out, C] dy_dtype; dx: [B, S, C] x_dtype; workspace: f32, pswin_ln_workspace(B * S, C) elements. */
int pswin_ln_gather_bwd(const void* dy, int dy_dtype, const int32_t* inv, const void* x, int x_dtype, const float* mean,
                        const float* rstd, const float* gamma, const float* dres, const float* res_scale,
                        float* dres_sum, void* dx, float* dgamma, float* dbeta, float* workspace, int B, int S, int n_out,
                        int C, void* stream);

/* pswin_ln_gather_bwd with a second output (round 4): ex[b][ex_map ? ex_map[t] : t] = bf16(ex_scale[b] * dx[b][t]) for every token, zero
 * rows at the ex_pads slots of every image -- the backward of the pswin_window_scatter_add that fed this LayerNorm's input (gather of the
 * residual-stream gradient into window order, or the plain bf16 cast of the Mlp branch), written while dx is in registers instead of by
 * pswin_window_gather re-reading it.  ex: bf16 [B, ex_rows, C]; ex_map: int32 [S] (token -> slot) or NULL (identity, ex_rows == S);
 * ex_pads: int32 [n_ex_pads = ex_rows - S] the slots no token maps to; ex_scale: f32 [B] or NULL.  x_dtype f32, C <= 1024.
 * ex == NULL: pswin_ln_gather_bwd. */
int pswin_ln_gather_bwd_ex(const void* dy, int dy_dtype, const int32_t* inv, const void* x, int x_dtype, const float* mean,
                           const float* rstd, const float* gamma, const float* dres, const float* res_scale, float* dres_sum, void* dx,
                           float* dgamma, float* dbeta, float* workspace, int B, int S, int n_out, int C, void* ex, const int32_t* ex_map,
                           int ex_rows, const float* ex_scale, const int32_t* ex_pads, int n_ex_pads, void* stream);

/* window_scatter_add followed at once by a token-order LayerNorm (the attention half of a block, HOT:516-536:
 * x1 = shortcut + DropPath(window_reverse(proj out) + proj bias), then norm2(x1)) in ONE pass:
 *   x1[b][t] = resid[b][t] + scale[b] * (win[b][inv[t]] + bias);   y[b][t] = LN(x1[b][t]) * gamma + beta
 * win: [B, n_slots, C] win_dtype; inv: int32 [S] or NULL (identity, n_slots == S); resid, x1: f32 [B, S, C]; scale: f32 [B]
 * or NULL; bias: f32 [C] or NULL; y: [B, S, C] y_dtype; mean, rstd: f32 [B, S].  Bitwise the result of
 * pswin_window_scatter_add + pswin_ln_gather_fwd; the backward pass is pswin_ln_gather_bwd + pswin_window_gather.
 * C % 8 == 0, C <= 1024. */
int pswin_scatter_add_ln_fwd(const void* win, int win_dtype, const int32_t* inv, const float* resid, const float* scale,
                             const float* bias, float* x1, const float* gamma, const float* beta, float eps, void* y,
                             int y_dtype, float* mean, float* rstd, int B, int S, int n_slots, int C, void* stream);

/* The same pass with the normalised rows written through a token -> slot map (round 4): y[b][out_map[t]] = LN(x1[b][t]) ..., zero rows at
 * the out_pads slots -- the residual add that ends a block (x + DropPath(mlp)) fused with the NEXT block's norm1 + shift + pad + window
 * partition (HOT:534-536 then HOT:503-513 of the following block): the new residual stream is written once and not re-read by a
 * LayerNorm + gather kernel.  y: [B, n_out, C]; out_map: int32 [S] or NULL (then n_out == S: pswin_scatter_add_ln_fwd);
 * out_pads: int32 [n_out_pads = n_out - S]. */
int pswin_scatter_add_ln_fwd_map(const void* win, int win_dtype, const int32_t* inv, const float* resid, const float* scale,
                                 const float* bias, float* x1, const float* gamma, const float* beta, float eps, void* y, int y_dtype,
                                 float* mean, float* rstd, int B, int S, int n_slots, int C, const int32_t* out_map, int n_out,
                                 const int32_t* out_pads, int n_out_pads, void* stream);

/* Output norms: y = LayerNorm(x) written channel-major, i.e. norm{i}(x).view(B, H, W, C).permute(0, 3, 1, 2).contiguous()
 * of HOT:975-977 in one pass (x: f32 [B, S, C]; y: f32 [B, C, S]; mean, rstd: f32 [B, S]), and its backward from the
 * NCHW gradient dy f32 [B, C, S] (dres as in pswin_ln_gather_bwd; workspace: pswin_ln_workspace(B * S, C) elements).
 * pswin_ln_nchw_supported(S, C) != 0 when the token count per image fits the kernels' tiling (S % (256 / lanes(C)) == 0). */
int pswin_ln_nchw_supported(int S, int C);
int pswin_ln_nchw_fwd(const float* x, const float* gamma, const float* beta, float eps, float* y, float* mean, float* rstd,
                      int B, int S, int C, void* stream);
/* The same with the residual add that closes the stage in front of it (HOT:536 followed by 975-977):
 *   x1[b][t] = x[b][t] + scale[b] * (win[b][t] + bias)   (win: bf16 [B, S, C] in token order, scale: f32 [B] or NULL, bias: f32 [C] or NULL),
 * written to x1 and normalised in the same pass; bitwise pswin_window_scatter_add followed by pswin_ln_nchw_fwd.  win == NULL: plain. */
int pswin_scatter_add_ln_nchw_fwd(const void* win_bf16, const float* x, const float* scale, const float* bias, float* x1,
                                  const float* gamma, const float* beta, float eps, float* y, float* mean, float* rstd, int B, int S,
                                  int C, void* stream);
int pswin_ln_nchw_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                      const float* dres, float* dx, float* dgamma, float* dbeta, float* workspace, int B, int S, int C,
                      void* stream);
/* The same with a second output: ex_bf16 [B, S, C] = bf16(ex_scale[b] * dx[b]) (ex_scale: f32 [B] DropPath factors or NULL = 1) -- the
 * gradient that flows into the MLP branch which closes the stage (x + DropPath(mlp(norm2(x))), HOT:536, followed by norm{i}, HOT:975):
 * the backward of pswin_window_scatter_add's cast, written in the pass that produces dx. */
int pswin_ln_nchw_bwd_ex(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                         const float* dres, float* dx, float* dgamma, float* dbeta, float* workspace, void* ex_bf16,
                         const float* ex_scale, int B, int S, int C, void* stream);

/* Workspace elements for the LayerNorm backward kernels over `rows` walked rows of width C. */
int pswin_ln_workspace(long long rows, int C);

/* The three LayerNorm backward entry points accept dgamma == dbeta == NULL ("partial rows only"): the kernel leaves
 * pswin_ln_partial_rows(rows, C) rows of [dgamma(C) | dbeta(C)] (or [dgamma | dbeta | dres_sum] when dres_sum != NULL;
 * dres_sum is then only a flag and is not written) in `workspace`, and the caller sums them later, typically together
 * with every other parameter-gradient reduction of the backward pass through pswin_reduce_jobs. */
int pswin_ln_partial_rows(long long rows, int C);

/* PatchMerging gather fused with its LayerNorm(4C) (HOT:563-574): y[b][i*W2+j] = LN(concat of the 4 tokens).
 * x: [B, H*W, C]; y: [B, H2*W2, 4C]; gamma, beta: f32 [4C]; mean, rstd: f32 [B, H2*W2]; C % 16 == 0, 4C <= 2048. */
int pswin_ln_patch_merge_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, float eps, void* y,
                             int y_dtype, float* mean, float* rstd, int B, int H, int W, int C, void* stream);

/* Its backward; dx: [B, H*W, C] (every token belongs to exactly one merged row);
 * workspace: pswin_ln_workspace(B * H2 * W2, 4 * C) elements. */
int pswin_ln_patch_merge_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean,
                             const float* rstd, const float* gamma, void* dx, float* dgamma, float* dbeta,
                             float* workspace, int B, int H, int W, int C, void* stream);

/* PatchMerging gather (HOT:560-573): out[b][i*W2+j][k*C + c] = x[b][(2i+dy_k)*W + 2j+dx_k][c] or 0 outside,
 * (dy,dx)_k = (0,0),(1,0),(0,1),(1,1); H2 = ceil(H/2), W2 = ceil(W/2).  x: [B, H*W, C], out: [B, H2*W2, 4C]. */
int pswin_patch_merge_gather(const void* x, int x_dtype, void* out, int out_dtype, int B, int H, int W, int C,
                             void* stream);

/* Its adjoint: dx[b][h*W+w][c] = dout[b][(h/2)*W2 + w/2][((h&1) + 2*(w&1))*C + c]. */
int pswin_patch_merge_scatter(const void* dout, int out_dtype, void* dx, int x_dtype, int B, int H, int W, int C,
                              void* stream);

/* BatchNorm2d + ReLU of the PatchEmbed stem (nn.BatchNorm2d + nn.ReLU after each 3x3 convolution, HOT:742-748) on a
 * channels-last activation viewed as rows y[M = N*H*W, C]:  z = relu((y - mean) * rstd * gamma + beta).
 * train != 0: mean / biased variance of the batch (nn.BatchNorm2d training semantics), running_mean / running_var
 * (may be NULL) updated in place with `momentum` and the unbiased variance; train == 0: the running statistics.
 * mean_offset (may be NULL): a per-channel constant the caller has NOT added to y although the reference adds it in
 * front of the BatchNorm (the convolution bias, HOT:743/746): it cancels in z and only shifts the tracked mean, so it is
 * folded into running_mean (train) / subtracted from it (eval) instead of costing a pass over the activation.
 * save_mean, save_rstd: f32 [C], kept for the backward pass.  y, z: dtype f32 or bf16, C % 8 == 0, C <= 1024.
 * workspace: f32, pswin_bn_workspace(C) elements. */
int pswin_bn_workspace(int C);
int pswin_bn_relu_fwd(const void* y, int dtype, const float* gamma, const float* beta, const float* mean_offset,
                      float eps, float momentum, int train, float* running_mean, float* running_var, void* z, float* save_mean, float* save_rstd,
                      float* workspace, long long M, int C, void* stream);

/* Its backward: g = dz * [z > 0];  dy = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)) (train) or
 * gamma * rstd * g (eval);  dgamma = sum g * xhat, dbeta = sum g (fixed summation order). */
int pswin_bn_relu_bwd(const void* dz, const void* y, int dtype, const float* gamma, const float* beta,
                      const float* save_mean, const float* save_rstd, int train, void* dy, float* dgamma, float* dbeta,
                      float* workspace, long long M, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused PatchEmbed stem (HOT:742-750: Conv3x3(3->32) BN ReLU Conv3x3(32->64) BN ReLU Conv4x4/s4(64->96)), bf16
 * operands / f32 accumulation, specialised for embed_dim 96, in_chans 3, patch_size 4.  Only y2 (the second
 * convolution's output) is ever stored at full resolution; see csrc/pswin_stem.hip for the data flow.
 * Packed operands (bf16): x4 [B][H][W][4] (channel 3 = 1), w1p [32][12][4] (tap-major, taps 9..11 / channel 3 zero),
 * w2p [9][64 out][32 in], w2t [9][32 in][64 out], w3p [16][96 out][64 in], w3t [16][64 in][96 out].
 * scale / shift: the BatchNorm folded to z = scale * y + shift per channel (f32).  workspace: f32,
 * pswin_stem_workspace(B, H, W) elements.
 * ---------------------------------------------------------------------------------------------- */
int pswin_stem_workspace(int B, int H, int W);
/* [B,3,H,W] f32 -> x4 */
int pswin_stem_pack_input(const float* x, int B, int H, int W, void* x4, void* stream);
/* sums: f32 [64 + 48*48]: per-channel sum / sum of squares of y1 = conv1(x) over all pixels (y1 is not stored), then
 * (want_xx) the input autocorrelation XX[k][k'] = sum_p xp[p][k] xp[p][k'] over the 48 tap-channel slots (slot
 * 4*4+3, the centre tap's ones channel, yields the plain sums and the pixel count) used by the backward pass. */
int pswin_stem_conv1_stats(const void* x4, const void* w1p, int B, int H, int W, int want_xx, float* sums,
                           float* workspace, void* stream);
/* y2 = conv2(relu(scale1 * conv1(x) + shift1)) (no bias), [B][H][W][64] bf16; sums2 (may be NULL): f32 [128] per-channel
 * sum / sum of squares of y2. */
int pswin_stem_conv2_fwd(const void* x4, const void* w1p, const float* scale1, const float* shift1, const void* w2p, int B,
                         int H, int W, void* y2, float* sums2, float* workspace, void* stream);
/* tokens[B*H/4*W/4][96] bf16 = conv3(relu(scale2 * y2 + shift2)) + bias3 */
int pswin_stem_conv3_fwd(const void* y2, const float* scale2, const float* shift2, const void* w3p, const float* bias3,
                         int B, int H, int W, void* tokens, void* stream);

/* Backward of the stem.  dtok: [M][96] bf16 gradient of the tokens.
 * conv3_bwd_stats: g2 = (dtok . W3)[pixel] * [scale2 y2 + shift2 > 0]; sums f32 [128] = per-channel sum g2 (= dbeta2),
 *   sum g2 * yhat2 (= dgamma2), yhat2 = a y2 + b.  prm: f32 [4][64] = scale2, shift2, a = rstd2, b = -mean2 rstd2.
 * conv3_bwd_data: dy2 = k1 g2 - P y2 - Q, [B][H][W][64] bf16.  prm: f32 [5][64] = scale2, shift2, k1, P, Q
 *   (training-mode BatchNorm backward: k1 = gamma rstd, P = k1 a mean(g2 yhat2), Q = k1 (mean g2 + b mean(g2 yhat2))).
 * conv3_wgrad: dw3 f32 [4 ky][8 waves][12][256] accumulator tiles (decoded by the host: stem.decode_dw3) of
 *   sum_tokens dtok (x) relu(scale2 y2 + shift2) patches.
 * conv2_wgrad: dw2 f32 [2][36][256] accumulator tiles (stem.decode_dw2) of sum_p dy2[p] (x) a1[p + tap], a1 recomputed.
 * perm (both; int32, may be NULL): where each accumulator element goes in the parameter layout ([96][64][4][4] /
 *   [64][32][3][3]); with it the final column sum writes the gradient directly in nn.Conv2d's weight layout.
 * conv2_bwd: out f32 [64 + 32*48]: sum g1 (= dbeta1), sum g1 * yhat1 (= dgamma1), G[ch][slot] = sum_p g1[p][ch] xp[p][slot]
 *   with g1 = conv2 data gradient of dy2 masked by relu(bn1 y1) > 0, never stored.  prm: f32 [4][32] = scale1, shift1,
 *   a = rstd1, b = -mean1 rstd1. */
int pswin_stem_conv3_bwd_stats(const void* dtok, const void* y2, const float* prm, const void* w3t, int B, int H, int W,
                               float* sums, float* workspace, void* stream);
int pswin_stem_conv3_bwd_data(const void* dtok, const void* y2, const float* prm, const void* w3t, int B, int H, int W,
                              void* dy2, void* stream);
int pswin_stem_conv3_wgrad(const void* dtok, const void* y2, const float* scale2, const float* shift2, int B, int H, int W,
                           const int32_t* perm, float* dw3, float* workspace, void* stream);
int pswin_stem_conv2_wgrad(const void* x4, const void* w1p, const float* scale1, const float* shift1, const void* dy2, int B,
                           int H, int W, const int32_t* perm, float* dw2, float* workspace, void* stream);
int pswin_stem_conv2_bwd(const void* x4, const void* w1p, const float* prm, const void* dy2, const void* w2t, int B, int H,
                         int W, float* out, float* workspace, void* stream);

/* Bias + exact GELU between fc1 and fc2 of Mlp (HOT:44-61).  y: [M, N] pre-activation WITHOUT the bias (the GEMM runs
 * without an epilogue), bias: f32 [N] or NULL, dtype f32 or bf16, N % 8 == 0.
 *   fwd: h = gelu(y + bias)      bwd: dy = dh * gelu'(y + bias), dbias[n] = sum_m dy[m][n] (f32, fixed order)
 * workspace: f32, pswin_bias_gelu_workspace(M, N) elements. */
int pswin_bias_gelu_fwd(const void* y, int dtype, const float* bias, void* h, long long M, int N, void* stream);
int pswin_bias_gelu_workspace(long long M, int N);
int pswin_bias_gelu_tune(int unr_fwd, int unr_bwd);   /* rows steps per block of the two kernels: 1, 2 or 4 (default 2, 4) */
int pswin_bias_gelu_partial_rows(long long M, int N, int dtype);   /* rows of N sums left in workspace when dbias == NULL */
int pswin_bias_gelu_bwd(const void* dh, const void* y, int dtype, const float* bias, void* dy, float* dbias,
                        float* workspace, long long M, int N, void* stream);

/* Small parameter kernels of the stem (one launch each).
 * pack_weights: nn.Conv2d weights f32 ([32,3,3,3], [64,32,3,3], [96,64,4,4]) -> w1p, w2p, w2t, w3p, w3t (bf16).
 * bn_fold: per-channel sum / sum of squares over `count` values of a bias-free convolution output -> prm f32 [4][C] =
 *   scale, shift, rstd, -mean * rstd (z = scale y + shift, yhat = rstd y - mean rstd); training != 0: batch statistics
 *   and nn.BatchNorm2d's running-statistics update (conv_bias, which cancels in z, is added to the tracked mean);
 *   training == 0: running statistics (conv_bias subtracted from the mean).
 * bn2_coefs: sums f32 [2][64] of conv3_bwd_stats + prm [4][64] -> prm5 f32 [5][64] = scale, shift, k1, P, Q for
 *   conv3_bwd_data (P = Q = 0 when training == 0).
 * conv1_wgrad: out5 of conv2_bwd, XX of conv1_stats (training), w1p, prm1 [4][32] -> dw1 f32 [32][3][3][3] and db1
 *   (zero in training: a bias in front of a BatchNorm has no gradient). */
int pswin_stem_pack_weights(const float* w1, const float* w2, const float* w3, void* w1p, void* w2p, void* w2t, void* w3p,
                            void* w3t, void* stream);
int pswin_stem_bn_fold(const float* sum, const float* sumsq, double count, const float* gamma, const float* beta,
                       const float* conv_bias, float eps, float momentum, int training, float* running_mean,
                       float* running_var, int C, float* prm, long long* num_batches_tracked, void* stream);
/* (num_batches_tracked: nn.BatchNorm2d's int64 counter, incremented by one on the device, or NULL) */
int pswin_stem_bn2_coefs(const float* sums, const float* prm, double count, int training, float* prm5, void* stream);
int pswin_stem_conv1_wgrad(const float* out5, const float* xx, const void* w1p, const float* prm1, double count, int training,
                           float* dw1, float* db1, void* stream);

/* Tiled bf16 GEMM for the Linear layers of stages 1-3 (qkv / proj / fc1 / fc2 / reduction, HOT:287, 309, 50-58, 575) and,
 * through a transposed copy of the weight, their data gradients:  y[M, N] = x[M, K] . w[N, K]^T (+ bias), bf16 in / out,
 * f32 accumulation, bias f32 [N] or NULL.  128 (or 64 / 96) x 192 macro tiles, both operands by LDS-DMA into XOR-swizzled
 * LDS tiles: double buffered with two workgroups per CU, or -- launches of at most 256 tiles -- four stages with three k-steps in
 * flight and one workgroup per CU (csrc/pswin_gemm_nt.hip).  Needs N % 192 == 0, K % 64 == 0, M >= 64 (pswin_gemm_nt_supported).
 * tile_m: 0 = choose, or 64 / 96 / 128 (96: plain epilogue only). */
/* The same product with a three-stage LDS ring (counted waits, one raw barrier per 64-row slab, one 8-wave workgroup per CU):
 * csrc/pswin_gemm_tn.hip, "Round 3".  One macro tile of 192 x 192 serves every Linear of the model (N % 192 == 0, K % 192 == 0).
 * partial: [splits, N, K] in `partial_dtype` (PSWIN_F32, or PSWIN_BF16 = each split's tile rounded once, as the library's batched GEMM
 * does for its row chunks); 1 <= splits <= M / 64; pswin_gemm_tn_ring_splits(M, N, K, target_wgs) suggests a split count for about
 * target_wgs workgroups (0 = 256: one per CU). */
int pswin_gemm_tn_ring_supported(long long M, int N, int K);
int pswin_gemm_tn_ring_splits(long long M, int N, int K, int target_wgs);
int pswin_gemm_tn_ring(const void* dy, const void* x, void* partial, int partial_dtype, long long M, int N, int K, int splits, void* stream);
/* The same launch with the bias gradient of the Linear riding along (autograd of HOT:287: db = column sums of dy): dbias_partial
 * f32 [splits][N] receives, per row split, the column sums of dy over the split's rows (summed by pswin_reduce_jobs like the
 * weight partials); columns zero_lo <= n < zero_hi are written as exact zeros (the K third of the qkv bias gradient sums to
 * zero analytically: ops.linear's zero_bias_cols).  dbias_partial = NULL: pswin_gemm_tn_ring. */
int pswin_gemm_tn_ring_bias(const void* dy, const void* x, void* partial, int partial_dtype, float* dbias_partial, int zero_lo, int zero_hi,
                            long long M, int N, int K, int splits, void* stream);

/* Several weight gradients in ONE launch per tile geometry (round 4).  A weight gradient is not needed before the backward pass ends,
 * so the host can queue them (ops.py: deferred weight gradients) and issue them together: with ~50 products in flight at once no
 * product needs a 256-way row split to fill the chip, so every workgroup contracts a few thousand rows and the partial-slab traffic
 * (256 x 192 x 192 x 2 B = 18.9 MB per product when each is launched alone) falls with the split count; per-launch ramps disappear.
 * Jobs are independent; each is the argument list of pswin_gemm_tn_ring_bias.  `jobs` is a HOST array (copied into the kernel
 * arguments, <= 48 jobs per launch and geometry; list the longest row ranges first).  splits == 1 with partial_dtype == PSWIN_F32 writes
 * the finished [N, K] gradient (no reduction needed). */
typedef struct pswin_tn_job {
    const void* dy;       /* bf16 [M, N] */
    const void* x;        /* bf16 [M, K] */
    void* partial;        /* [splits, N, K] in partial_dtype */
    float* dbias_partial; /* f32 [splits, N] or NULL */
    long long M;
    int N, K, splits, partial_dtype, zero_lo, zero_hi;
} pswin_tn_job;
int pswin_gemm_tn_ring_jobs(const pswin_tn_job* jobs, int n_jobs, void* stream);

typedef struct pswin_transpose_job {
    const void* src; /* bf16 [rows][cols] */
    void* dst;       /* bf16 [cols][rows] */
    int rows;
    int cols;
} pswin_transpose_job;
/* dst = src^T for every job in one launch (rows, cols multiples of 64): the per-step [K][N] bf16 copies of the Linear
 * weights with which pswin_gemm_nt computes data gradients.  `jobs` is a HOST array, copied into the kernel arguments. */
int pswin_transpose_jobs(const pswin_transpose_job* jobs, int n_jobs, void* stream);

/* AdamW over the one flat fp32 parameter buffer of a model: the update that ends every training step of the path (the reference runs
 * torch.optim.AdamW through mmcv's OptimizerHook, mmdet/apis/train.py:91-112 with the configs/swin files: lr 1e-4, betas (0.9, 0.999),
 * weight_decay 0.05), fused with the bf16 copy of the updated parameters that the next step's kernels read.  p, g, m, v: f32 [n]
 * (n a multiple of 4, 16-byte aligned); p_bf16: bf16 [n] or NULL; step: DEVICE pointer to the number (>= 1, as a float) of the step being
 * taken.  Same arithmetic, element by element, as torch.optim.AdamW. */
int pswin_adamw_flat(float* p, const float* g, float* m, float* v, void* p_bf16, long long n, double lr, double beta1, double beta2, double eps,
                     double weight_decay, const float* step, void* stream);
/* The same update with PARAMETER GROUPS, as the reference's optimizer is configured
 * (configs/swin/mask_rcnn_swin_tiny_patch4_window7_mstrain_480-800_adamw_1x_coco.py:64-67: paramwise_cfg.custom_keys give every parameter
 * whose name contains 'norm' / 'relative_position_bias_table' / 'absolute_pos_embed' decay_mult = 0; mmcv's optimizer constructor turns that
 * into per-parameter groups).  group_of: DEVICE array of n / 4 bytes, the group (< n_groups <= PSWIN_ADAMW_MAX_GROUPS) of elements
 * 4 i .. 4 i + 3 (parameters start on 64-element boundaries of the flat buffer, so a granule never straddles two); lr_mult / decay_mult:
 * HOST arrays of n_groups multipliers (copied into the kernel arguments).  Group k is updated exactly as torch.optim.AdamW updates a
 * group with lr * lr_mult[k] and weight_decay * decay_mult[k].  group_of == NULL (n_groups 0): one group, = pswin_adamw_flat. */
#define PSWIN_ADAMW_MAX_GROUPS 8
int pswin_adamw_flat_groups(float* p, const float* g, float* m, float* v, void* p_bf16, long long n, const unsigned char* group_of,
                            int n_groups, const float* lr_mult, const float* decay_mult, double lr, double beta1, double beta2, double eps,
                            double weight_decay, const float* step, void* stream);
/* ---- the caller's side of the path: RoIAlign over an FPN pyramid ------------------------------------------------
 * configs/_base_/models/mask_rcnn_swin_fpn.py:44-48, 63-67 (SingleRoIExtractor, RoIAlign output 7 / 14, sampling_ratio = 0, strides
 * 4..32) and mmdet/models/roi_heads/roi_extractors/single_level_roi_extractor.py:78-108.  RoIAlign itself is mmcv.ops (not in the
 * reference tree): the published operator is implemented (csrc/pswin_roi.hip states it); parity with the reference is unpinned.
 * Feature maps are NHWC [B, H, W, C] (`dtype`), rois f32 [R, 5] = (batch index, x1, y1, x2, y2) in image pixels, roi_level int32 [R]
 * = the pyramid level of every RoI (map_roi_levels, :55-60, computed by the caller), out / dout [R, P, P, C] (`dtype`).
 * Backward ACCUMULATES (f32 atomics) into the f32 NHWC maps levels->dfeat, which the caller zeroes. */
#define PSWIN_ROI_MAX_LEVELS 4
typedef struct pswin_roi_levels {
    const void* feat[PSWIN_ROI_MAX_LEVELS]; /* forward: feature maps (NULL in backward) */
    float* dfeat[PSWIN_ROI_MAX_LEVELS];     /* backward: gradient maps (NULL in forward) */
    int H[PSWIN_ROI_MAX_LEVELS];
    int W[PSWIN_ROI_MAX_LEVELS];
    float spatial_scale[PSWIN_ROI_MAX_LEVELS]; /* 1 / stride */
    int n_levels;
} pswin_roi_levels;
int pswin_roi_align_supported(int C, int dtype);
int pswin_roi_align_fwd(const pswin_roi_levels* levels, const float* rois, const int32_t* roi_level, int R, int C, int P, int sampling_ratio,
                        int aligned, int dtype, void* out, void* stream);
int pswin_roi_align_bwd(const pswin_roi_levels* levels, const float* rois, const int32_t* roi_level, int R, int C, int P, int sampling_ratio,
                        int aligned, int dtype, const void* dout, void* stream);

/* Greedy NMS over `groups` independent lists of boxes sorted by descending score -- the proposal stage of RPNHead (mmcv.ops.nms inside
 * mmdet/models/dense_heads/rpn_head.py: one list per image and pyramid level): keep[i] = no kept j < i with IoU(i, j) > iou_threshold.
 * boxes f32 [groups][nmax][4] (x1, y1, x2, y2), counts int32 [groups] (valid boxes per group) or NULL (= nmax everywhere), keep uint8
 * [groups][nmax] (1 = kept, 0 = suppressed or past the group's count).  nmax <= 2048, a multiple of 64.  Unpinned like RoIAlign (mmcv.ops is not in
 * the reference tree): tests check it against the sequential rule. */
int pswin_nms_workspace(int groups, int nmax);                 /* bytes of workspace for pswin_nms_groups (the suppression bit masks); groups <= 2048 */
int pswin_nms_groups(const float* boxes, const int32_t* counts, int groups, int nmax, float iou_threshold, unsigned char* keep, void* workspace,
                     void* stream);

int pswin_gemm_nt_supported(long long M, int K, int N);
int pswin_gemm_nt(const void* x, const void* w, const float* bias, void* y, long long M, int K, int N, int tile_m, void* stream);
/* The same product with an f32 result (y: f32 [M, N], M * N * 4 < 4 GiB): PatchMerging.reduction (HOT:575), whose output is the fp32
 * residual stream of the next stage -- written once instead of as bf16 plus a cast. */
int pswin_gemm_nt_f32(const void* x, const void* w, const float* bias, float* y, long long M, int K, int N, int tile_m, void* stream);
/* The data gradient of the Mlp's fc2 fused with the backward of fc1's bias + nn.GELU (HOT:50-58):
 *   dpre[M, N] = (dy[M, K] . w_t[N, K]^T) * gelu'(pre + bias),   partial[t][n] = sum over the rows of row tile t of dpre[., n]
 * with w_t = fc2.weight^T ([hidden N, K]), pre = the pre-activation fc1(x) without bias, bias = fc1.bias (f32 [N] or NULL).
 * dL/dh is never written.  partial: f32 [pswin_gemm_nt_partial_rows(M, tile_m), N]; its column sums are the fc1 bias
 * gradient (pswin_reduce_jobs).  tile_m: 64 or 128. */
/* fc1 of the Mlp with its bias + nn.GELU in the GEMM epilogue (HOT:50-57):  pre = x . w^T (no bias; kept for the backward pass),
 * h = gelu(pre + bias), both bf16 [M, N], bias f32 [N].  h is computed from the bf16-rounded pre: bit-identical to
 * pswin_gemm_nt followed by pswin_bias_gelu_fwd, one pass over [M, N] less.  tile_m: 64 or 128. */
int pswin_gemm_nt_gelu_fwd(const void* x, const void* w, const float* bias, void* pre, void* h, long long M, int K, int N, int tile_m,
                           void* stream);
int pswin_gemm_nt_partial_rows(long long M, int tile_m);
int pswin_gemm_nt_gelu_bwd(const void* dy, const void* w_t, const void* pre, const float* bias, void* dpre, float* partial, long long M,
                           int K, int N, int tile_m, void* stream);

/* Streaming GEMM for the Linear layers of the high-resolution stages (qkv / proj / fc1 / fc2 of stage 0, HOT:287, 309,
 * 50-58; proj of stage 1):  y[M, N] = x[M, K] . W^T (+ bias), bf16 in / out, f32 accumulation, the whole weight resident
 * in LDS.  transpose_w == 0: w is [N, K] (nn.Linear layout, forward pass); transpose_w != 0: w is [K, N], i.e. the
 * data gradient dx[M, K'] = dy[M, N'] . W with w = the same nn.Linear weight [N', K'] passed as K = N', N = K'.
 * bias: f32 [N] or NULL.  pswin_gemm_skinny_supported(K, N) != 0 for the (K, N) pairs the kernel is instantiated for. */
int pswin_gemm_skinny_supported(int K, int N);
int pswin_gemm_skinny(const void* x, const void* w, const float* bias, void* y, long long M, int K, int N, int transpose_w,
                      void* stream);

/* fc1 + bias + GELU of Mlp (HOT:50-57) fused into the streaming GEMM for the stage-0 shape (K = 96, N = 384):
 *   fwd: h[M, N] = gelu(x[M, K] . W^T + bias)                        -- the pre-activation is never stored
 *   bwd: dy = dh * gelu'(x . W^T + bias) with the pre-activation RECOMPUTED, dbias[n] = sum_m dy[m][n]
 * (dy is then the output gradient of the fc1 GEMM: dx = dy . W and dW = dy^T x follow as for any Linear).
 * x, w, h, dh, dy: bf16; bias, dbias: f32; workspace: f32, pswin_fc1_gelu_workspace(N) elements. */
int pswin_fc1_gelu_supported(int K, int N);
int pswin_fc1_gelu_fwd(const void* x, const void* w, const float* bias, void* h, long long M, int K, int N, void* stream);
int pswin_fc1_gelu_workspace(int N);
int pswin_fc1_gelu_partial_rows(long long M);                      /* rows of N sums left in workspace when dbias == NULL */
int pswin_fc1_gelu_bwd(const void* x, const void* w, const float* bias, const void* dh, void* dy, float* dbias,
                       float* workspace, long long M, int K, int N, void* stream);

/* Stage-0 Mlp backward, first half, as ONE pass (round 3): g = (dy . W2) * gelu'(x . W1^T + b1) -- fc2's data gradient (HOT:58), the
 * backward of nn.GELU (HOT:57) on the recomputed fc1 pre-activation (HOT:56) and the fc1 bias gradient's partial sums; dh = dy . W2
 * (201 MB per block at batch 8) is never written.  x, dy [M, C] bf16 rows; w1 = fc1.weight [hidden, C], w2 = fc2.weight [C, hidden]
 * bf16; b1 f32 [hidden] or NULL; g [M, hidden] bf16; workspace f32 [pswin_mlp0_bwd_partial_rows(M)][hidden] receives the
 * per-workgroup column sums of g; dbias1 f32 [hidden] their fixed-order sum, or NULL (partial rows only); workspace = NULL: no column
 * sums (a caller that runs fc1's weight gradient on pswin_gemm_tn_ring_bias gets them from that launch).  C = 96, hidden = 384. */
/* Stage-0 Mlp forward in one pass (HOT:50-58 without fc2's bias, which the caller adds with the residual): h[M, hidden] = gelu(x W1^T +
 * b1) is written once for the backward pass and y[M, C] = h W2^T is formed from the first product's accumulators (h is not read
 * back).  Same shapes and dtypes as pswin_mlp0_bwd; pswin_mlp0_bwd_supported(C, hidden) says whether both exist. */
int pswin_mlp0_fwd(const void* x, const void* w1, const float* b1, const void* w2, void* h, void* y, long long M, int C, int hidden, void* stream);
int pswin_mlp0_bwd_supported(int C, int hidden);
int pswin_mlp0_bwd_partial_rows(long long M);
int pswin_mlp0_bwd(const void* x, const void* w1, const float* b1, const void* dy, const void* w2, void* g, float* dbias1, float* workspace,
                   long long M, int C, int hidden, void* stream);

/* Column sums of a row-major [M, N] matrix in fp32: out[n] = sum_m x[m][n] (fixed summation order).  The bias
 * gradient of every Linear on the path (autograd of nn.Linear, HOT:50-52, 236, 323) and the reduction of split-K
 * weight-gradient partials.  N % 8 == 0; workspace: f32, pswin_colsum_workspace(M, N, dtype) elements. */
int pswin_colsum_workspace(long long M, int N, int dtype);
int pswin_colsum(const void* x, int dtype, long long M, int N, float* out, float* workspace, void* stream);
/* out == NULL: first stage only; pswin_colsum_workspace(M, N, dtype) / N partial rows of N sums stay in workspace. */
/* First stage only, for a matrix whose columns [skip_lo, skip_hi) are known to sum to zero: they are not read and their
 * partial sums are zeros.  Used for the qkv bias gradient (autograd of HOT:236, 287): the K third of d(qkv) sums to zero
 * over the window tokens analytically (dK = dS^T Q and every row of dS sums to 0 because softmax rows sum to 1), so a
 * third of the largest gradient tensor of each block is not re-read.  skip_lo, skip_hi: multiples of 8 (bf16) / 4 (f32). */
int pswin_colsum_skip(const void* x, int dtype, long long M, int N, int skip_lo, int skip_hi, float* workspace, void* stream);

/* Grouped column sums: dst[c] = sum_{r < rows} src[r * ld + c], c < cols, for up to a few hundred independent jobs in
 * ONE launch per 96 jobs (fixed summation order, bitwise reproducible).  The autograd of the path produces ~120 small
 * parameter-gradient reductions per backward pass (split-K partials of dW = dY^T X, bias-gradient partial rows, the
 * LayerNorm dgamma / dbeta rows; reference: torch autograd of HOT:50-58, 236, 323, 503, 512); none of them is needed
 * before the pass ends, so the host queues them and issues them here together.  `jobs` is a HOST array (it is copied
 * into the kernel arguments, so it may be freed on return and a captured hipGraph keeps its own copy).
 * src: f32 or bf16 (dtype), 16-byte aligned, ld % 8 == 0 (bf16) / % 4 == 0 (f32); cols likewise; dst: f32, aligned. */
typedef struct pswin_reduce_job {
    const void* src;
    float* dst;
    int dtype;
    int rows;
    int cols;
    int ld;
} pswin_reduce_job;
int pswin_reduce_jobs(const pswin_reduce_job* jobs, int n_jobs, void* stream);

/* Static 4-tap row interpolation (the two F.grid_sample calls of PitchAttentionModule.get_rotated,
 * HOT:1038, 1090, with input-independent grids; lzx/pano_rotate.py:169-187):
 * out[b][p][:] = sum_k wgt[p][k] * x[b][idx[p][k]][:]   x: [B, S, C] f32, out: [B, P, C] f32,
 * idx: int32 [P, 4], wgt: f32 [P, 4]. */
int pswin_interp_rows(const float* x, const int32_t* idx, const float* wgt, float* out, int B, int S, int P, int C,
                      void* stream);

/* Its adjoint (atomic f32 adds): dx[b][idx[p][k]][:] += wgt[p][k] * dout[b][p][:]; dx must be zeroed by the
 * caller. */
int pswin_interp_rows_adjoint(const float* dout, const int32_t* idx, const float* wgt, float* dx, int B, int S,
                              int P, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Window attention (the hot kernel): BasicWindowAttention.forward (HOT:274-311) between the qkv and
 * proj Linear layers, and PitchAttentionModule._attention (HOT:1206-1237) between q/k/v_linear and proj.
 * ---------------------------------------------------------------------------------------------- */

/* Score bias of window n, head h (BasicWindowAttention._sphere_bias, HOT:241-272, plus the shifted-window mask add
 * of HOT:295-303), evaluated inside the attention kernels once per (bias window, head) and shared by every image
 * of the batch (window n uses bias window n % n_bias_windows):
 *   bias(i,j) = (dist ? dist[wb % n_dist](i,j) * alpha[idx(i,j)][h] : 0) + beta[idx(i,j)][h]
 *               + (mask ? mask[wb % n_mask](i,j) : 0)
 *   idx(i,j)  = (i/7 - j/7 + 6) * 13 + (i%7 - j%7 + 6)         (make_relative_position_index, HOT:95-129)
 * dist / mask are passed as zero-padded 64 x 64 TILES (pswin_attn_pad_tiles): the forward kernel reads tile[i][j],
 * the backward kernel reads the TRANSPOSED tile[j][i] (for the symmetric self-attention tables both are the same
 * buffer).  alpha (ignored when dist is NULL: planar mode, HOT:257-258), beta: f32 [169, heads].
 * n_bias_windows must be a multiple of n_dist and of n_mask. */

/* [n, 49, 49] -> zero-padded [n, 64, 64] tiles, transposed when `transpose` != 0. */
int pswin_attn_pad_tiles(const float* src, int n, int transpose, float* dst, void* stream);

/* out[n][i][h*32 + d] = sum_j softmax_j(scale * q[n][i][h][:] . k[n][j][h][:] + bias(i, j)) v[n][j][h][d]
 * q, k, v: element pointers to head 0 of token 0 of window 0; token rows are ld_qkv elements apart, heads 32
 * elements apart (one fused [n*49, 3C] qkv buffer: q = base, k = base + C, v = base + 2C, ld_qkv = 3C).
 * out: [n_windows*49, ld_out]; lse: f32 [n_windows, heads, 64] = log-sum-exp of every score row (+inf in rows
 * >= 49), kept for the backward pass.  n_windows % n_bias_windows == 0.
 * n_chunks: how many work items share one bias window's batch loop (0 = pswin_attn_suggest_chunks; otherwise a
 * divisor of n_windows / n_bias_windows).
 * dtype: q, k, v, out all PSWIN_F32 (exact-f32 MFMA) or all PSWIN_BF16 (bf16 MFMA, f32 softmax/accumulate).
 * Pointers 16-byte aligned, ld_qkv % 8 == 0, ld_out % 8 == 0. */
int pswin_attn_fwd(const void* q, const void* k, const void* v, int ld_qkv, const float* dist_tiles, int n_dist,
                   const float* alpha, const float* beta, const float* mask_tiles, int n_mask, void* out, int ld_out,
                   float* lse, int n_chunks, int n_windows, int n_bias_windows, int heads, float scale, int dtype,
                   void* stream);

/* Host helper: the number of batch-loop chunks the library would pick for this geometry (enough independent
 * waves to fill the chip while keeping the per-work-item bias reuse long), for the forward (backward = 0) or the
 * backward (backward = 1) kernel.  Returns the chunk count (>= 1) or PSWIN_ERR_ARG. */
int pswin_attn_suggest_chunks(int n_windows, int n_bias_windows, int heads, int backward);

/* The whole WindowAttention module per window in ONE kernel (HOT:274-323: self.qkv, the attention core above, self.proj):
 *   y = softmax(scale * q k^T + bias) v @ Wproj^T   with [q | k | v] = x @ Wqkv^T + b_qkv,
 * x: [n_windows*49, C] window rows (the output of pswin_ln_gather_fwd), y: [n_windows*49, C] WITHOUT the proj bias (the
 * residual scatter kernel adds it, as on the unfused path).  The qkv tensor never exists in HBM: both weight matrices
 * are resident in LDS as MFMA operand fragments and every product of a window runs out of registers
 * (csrc/pswin_fused.hip).  w_qkv: [3C, C], w_proj: [C, C] (nn.Linear layout), b_qkv: f32 [3C] or NULL; dist / mask tiles,
 * tables, n_bias_windows and scale as pswin_attn_fwd (forward tiles, not transposed).
 * Training: pass qkv_out (bf16 [n, heads, 3, 49, 32]: q, k, v of a head as contiguous blocks, the packing pswin_attn_bwd_ex
 * reads), att_out [n*49, C] (the attention output before proj, row-major: the proj weight-gradient GEMM reads it) and
 * lse_out f32 [n, heads, 64] (all three or none).
 * Specialised for C = 96, heads = 3, PSWIN_BF16 (PanoSwin-T / -S stage 0); pswin_win_attn_fused_supported says so,
 * anything else returns PSWIN_ERR_UNSUPPORTED.  Rounding points are those of the unfused bf16 path (qkv and the attention
 * output rounded to bf16, f32 scores / softmax / accumulation). */
int pswin_win_attn_fused_supported(int C, int heads, int dtype);
int pswin_win_attn_fused_fwd(const void* x, const void* w_qkv, const float* b_qkv, const void* w_proj, const float* dist_tiles,
                             int n_dist, const float* alpha, const float* beta, const float* mask_tiles, int n_mask, void* y,
                             void* qkv_out, void* att_out, float* lse_out, long long n_windows, int n_bias_windows, int C,
                             int heads, float scale, int dtype, void* stream);

/* The same fusion WITHOUT the proj Linear for the stages behind the first one (csrc/pswin_qkvattn.hip, round 3): qkv Linear (HOT:287) +
 * attention core (HOT:288-308) of one (window, head) per wave, C = 192 (6 heads) or 384 (12); the head's 96 weight rows stay in LDS for
 * the lifetime of a workgroup.  y: attention output rows [n_windows * 49, C] (the input of the proj GEMM).  qkv_out / lse_out: both or
 * neither (training): q, k, v as packed [n][heads][3][49][32] blocks and the log-sum-exp rows [n][heads][64], what pswin_attn_bwd_ex
 * (ld_in 32, win_stride heads * 3 * 49 * 32, head_stride 3 * 49 * 32) reads.  Other arguments as pswin_win_attn_fused_fwd. */
int pswin_qkv_attn_fused_supported(int C, int heads, int dtype);
int pswin_qkv_attn_fused_fwd(const void* x, const void* w_qkv, const float* b_qkv, const float* dist_tiles, int n_dist, const float* alpha,
                             const float* beta, const float* mask_tiles, int n_mask, void* y, void* qkv_out, float* lse_out, long long n_windows,
                             int n_bias_windows, int C, int heads, float scale, int dtype, void* stream);

/* Gradients of pswin_attn_fwd.  dq, dk, dv use the q/k/v addressing (ld_dqkv), dout the out addressing.
 * dist_tiles_t / mask_tiles_t: the TRANSPOSED tiles.  n_chunks splits the batch loop (1 <= n_chunks <=
 * n_windows / n_bias_windows, must divide it).
 * dscore_sum: f32 [n_chunks * n_bias_windows, heads, 64(j), 64(i)] or NULL: per work item, the sum over its images
 * of dScore (= the gradient w.r.t. the bias), transposed like the backward tiles; input of pswin_attn_table_grads. */
int pswin_attn_bwd(const void* q, const void* k, const void* v, int ld_qkv, const float* dist_tiles_t, int n_dist,
                   const float* alpha, const float* beta, const float* mask_tiles_t, int n_mask, const void* dout,
                   int ld_out, const float* lse, void* dq, void* dk, void* dv, int ld_dqkv, float* dscore_sum,
                   int n_chunks, int n_windows, int n_bias_windows, int heads, float scale, int dtype, void* stream);
/* The same with q / k / v in any packing: window w, head h starts w * qkv_window_stride + h * qkv_head_stride elements behind the
 * q / k / v pointer, token rows ld_qkv elements apart.  pswin_win_attn_fused_fwd saves q, k, v as contiguous [49][32] blocks
 * ([n][heads][3][49][32]: ld_qkv = 32, head stride 3 * 49 * 32, window stride heads * 3 * 49 * 32; k = q + 49 * 32 elements,
 * v = q + 2 * 49 * 32), which it can write -- and this kernel read -- as whole 1 KB runs instead of 64-byte row segments. */
int pswin_attn_bwd_ex(const void* q, const void* k, const void* v, int ld_qkv, long long qkv_window_stride, int qkv_head_stride,
                      const float* dist_tiles_t, int n_dist, const float* alpha, const float* beta, const float* mask_tiles_t,
                      int n_mask, const void* dout, int ld_out, const float* lse, void* dq, void* dk, void* dv, int ld_dqkv,
                      float* dscore_sum, int n_chunks, int n_windows, int n_bias_windows, int heads, float scale, int dtype,
                      void* stream);


/* Table gradients (adjoint of the bias w.r.t. alpha, beta), in a fixed summation order:
 *   dbeta[t][h] = sum over tiles and (i,j) with idx(i,j) = t of g ;  dalpha[t][h] = the same sum of g * dist(i,j).
 * dscore_sum: the n_tiles = n_chunks * n_bias_windows tiles written by pswin_attn_bwd; dist_tiles_t as there (NULL in
 * planar mode: dalpha untouched); dalpha, dbeta: f32 [169, heads], overwritten.
 * workspace: f32, pswin_attn_table_grads_workspace(heads) elements. */
int pswin_attn_table_grads_workspace(int heads);
int pswin_attn_table_grads_partial_rows(int n_tiles, int heads);
int pswin_attn_table_grads(const float* dscore_sum, int n_tiles, int n_bias_windows, const float* dist_tiles_t,
                           int n_dist, int heads, float* dalpha, float* dbeta, float* workspace, void* stream);

/* The same for several attention modules at once.  stages: 1 = only the partial sums over the dScore tiles (best issued
 * right after pswin_attn_bwd while the tiles are still in the last-level cache), 2 = only the sum of those partials
 * and the per-bin sums (one binning launch for all jobs), 3 = both, 4 = only the per-bin sums: the caller has summed
 * the partial rows itself, e.g. as pswin_reduce_jobs jobs together with the other parameter-gradient reductions of
 * the backward pass.  Workspace layout for that: pswin_attn_table_grads_partial_rows(n_tiles, heads) rows of
 * ld = pswin_attn_table_grads_workspace(heads) / 129 floats from the start, their sum (ld floats) at offset 128 * ld.  `jobs` is a HOST array; it is copied into the kernel
 * arguments (32 jobs per launch).  Field meaning as the arguments of pswin_attn_table_grads. */
typedef struct pswin_table_grad_job {
    const float* dscore_sum;
    const float* dist_tiles_t;
    float* dalpha;
    float* dbeta;
    float* workspace;
    int n_tiles;
    int n_bias_windows;
    int n_dist;
    int heads;
} pswin_table_grad_job;
int pswin_attn_table_grads_batch(const pswin_table_grad_job* jobs, int n_jobs, int stages, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PSWIN_H_ */
