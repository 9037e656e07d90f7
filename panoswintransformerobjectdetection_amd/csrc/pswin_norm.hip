// LayerNorm fused with the PanoSwin row movers (gfx950).
//
// In the reference every block does  norm1 -> cat uv -> roll/flip/cat -> pad -> window_partition  (HOT:503-513,
// 64-75) and PatchMerging does  pad -> 4 strided slices -> cat -> norm  (HOT:563-574): a LayerNorm pass plus 4-7
// full-tensor copies.  Here the normalisation happens inside the indexed row copy: one read of the source rows,
// one write of the normalised rows in their destination layout (window slots / merged tokens), statistics kept
// per source token for the backward pass.  The same kernel with an identity map is the plain LayerNorm used for
// norm2 and the output norms, writing bf16 directly when the consumer is a bf16 GEMM.
// HOT = mmdet/models/backbones/simple_panoswin_transformer.py of the reference.
//
// Mapping: a row of C elements is L lanes x up to 4 chunks of 4 elements (L = 2..64, a power of two, chosen on
// the host so that L * 16 >= C); consecutive lanes read consecutive 16-byte chunks; row reductions are log2(L)
// __shfl_xor steps.  Statistics and all arithmetic are fp32 regardless of the I/O dtype.
#include "pswin_common.hpp"

using namespace pswin;

namespace {

constexpr int THREADS = 256;

// Sum over the L lanes of a row (every lane receives it).  Steps inside a 16-lane DPP row run on the VALU
// (v_add_f32 with a DPP modifier, ~5 cycles): __shfl_xor compiles to ds_bpermute_b32, an LDS-crossbar round trip of
// ~100+ cycles each, and the 2 x log2(L) dependent ones per row were a quarter of a LayerNorm backward iteration.
template <int CTRL>
__device__ inline float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int L>
__device__ inline float row_sum(float v) {
    if constexpr (L >= 2) v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]: lane ^ 1
    if constexpr (L >= 4) v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]: lane ^ 2
    if constexpr (L >= 8) v = dpp_add<0x141>(v);     // row_half_mirror: the other quad of the 8 (quads are uniform by now)
    if constexpr (L >= 16) v = dpp_add<0x140>(v);    // row_mirror: the other half of the 16
    if constexpr (L >= 32) v += __shfl_xor(v, 16);
    if constexpr (L >= 64) v += __shfl_xor(v, 32);
    return v;
}

// MODE 0: out row (b, slot) <- source token map[slot] (or slot when map == nullptr), -1 = zero row.
// MODE 1: PatchMerging: out row (b, i*W2 + j) <- concat of the 4 tokens (2i+dy, 2j+dx), zeros outside (H, W).
struct RowSrc {
    int mode;
    const int32_t* map;   // MODE 0
    int S, n_out;         // tokens per image, out rows per image
    int H, W, W2;         // MODE 1
};

// source element offset (in elements, within the image) of chunk `ch` of out row r; -1 = zero
template <int MODE>
__device__ inline long long src_offset(const RowSrc& rs, int r, int ch, int C, int mode0_src) {
    if constexpr (MODE == 0) {
        return mode0_src < 0 ? -1 : (long long)mode0_src * C + 4 * ch;
    } else {
        const int cq = C / 4;                 // C is the OUTPUT width (4 * input channels): chunks per quarter = cq / 4
        const int per_q = cq / 4;
        const int k = ch / per_q, within = ch - k * per_q;
        const int i = r / rs.W2, j = r - i * rs.W2;
        const int yy = 2 * i + (k & 1), xx = 2 * j + (k >> 1);
        if (yy >= rs.H || xx >= rs.W) return -1;
        return ((long long)yy * rs.W + xx) * (C / 4) + 4 * within;
    }
}

template <int MODE, int XDT, int YDT, int L, int NCH>
__global__ __launch_bounds__(THREADS) void ln_fwd_kernel(const void* __restrict__ x, RowSrc rs,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps,
                                                         void* __restrict__ y, float* __restrict__ mean,
                                                         float* __restrict__ rstd, long long rows, int C,
                                                         const float* __restrict__ add) {
    constexpr int RPB = THREADS / L;
    const int lane = threadIdx.x % L;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / L;
    if (row >= rows) return;
    const int b = (int)(row / rs.n_out);
    const int r = (int)(row - (long long)b * rs.n_out);
    const int nchunks = C / 4;
    const size_t img_in = (size_t)b * rs.S * (MODE == 0 ? C : C / 4);
    int src0 = 0;
    if constexpr (MODE == 0) src0 = rs.map ? rs.map[r] : r;
    f32x4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = lane + k * L;
        v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ch < nchunks) {
            const long long off = src_offset<MODE>(rs, r, ch, C, src0);
            if (off >= 0) v[k] = load4<XDT>(x, img_in + (size_t)off);
            s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
        }
    }
    if (MODE == 0 && src0 < 0) {            // zero (padding) slot: stays zero, as F.pad after norm1 does (HOT:504-512)
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + k * L;
            if (ch < nchunks) store4<YDT>(y, (size_t)row * C + 4 * (size_t)ch, f32x4{0.f, 0.f, 0.f, 0.f});
        }
        return;
    }
    const float mu = row_sum<L>(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = lane + k * L;
        if (ch < nchunks) {
            const f32x4 d = v[k] - mu;
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
    const float rs_ = rsqrtf(row_sum<L>(q) / (float)C + eps);
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = lane + k * L;
        if (ch < nchunks) {
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + 4 * ch);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(beta + 4 * ch);
            f32x4 o = (v[k] - mu) * rs_ * g4 + b4;
            // add: one more f32 row per output row of an image ([n_out, C]: the absolute position encoding, HOT:932-934, added to the
            // normalised patch embedding while it is in registers instead of by a pass over the residual stream)
            if (add) o = o + *reinterpret_cast<const f32x4*>(add + (size_t)r * C + 4 * (size_t)ch);
            store4<YDT>(y, (size_t)row * C + 4 * (size_t)ch, o);
        }
    }
    if (lane == 0) {
        // statistics live at the SOURCE token for MODE 0 (the backward pass walks tokens), at the out row for MODE 1
        const size_t at = (MODE == 0) ? (size_t)b * rs.S + src0 : (size_t)row;
        mean[at] = mu;
        rstd[at] = rs_;
    }
}

// optional destination map of ln_add_fwd_kernel's normalised rows: token t of image b -> y row b * n_out + map[t]; pads = the slots of an
// image no token maps to (zero rows).  map == nullptr: token order (row = b * S + t).
struct OutMap {
    const int32_t* map;
    const int32_t* pads;
    int n_out, n_pads, B;
};
// optional second output of ln_bwd_kernel (MODE 0): ex[b][map ? map[t] : t] = bf16(scale[b] * dx[b][t]), zero rows at the pad slots --
// the window gather (or plain cast) that turns the residual-stream gradient into the branch gradient the next backward kernel reads
// (window_scatter_add's backward), written while dx is in registers instead of by a pass of its own
struct BwdExtra {
    void* ex;
    const int32_t* map;
    const float* scale;
    const int32_t* pads;
    int n_rows, n_pads, B;
};

// window_scatter_add + LayerNorm in token order, one pass (the attention half of a block ends with
// x1 = x + DropPath(window_reverse(proj(.)) + b) and norm2(x1) follows at once, HOT:516-536):
//   x1[b][t] = resid[b][t] + scale[b] * (win[b][inv[t]] + bias);   y[b][t] = LN(x1[b][t]) * gamma + beta
// x1 is written for the shortcut and the backward pass, but not read back by a separate LayerNorm kernel.  The
// arithmetic (and its order) is that of window_scatter_add_kernel followed by ln_fwd_kernel: bitwise the same result.
template <int WDT, int YDT, int L, int NCH>
__global__ __launch_bounds__(THREADS) void ln_add_fwd_kernel(const void* __restrict__ win, const int32_t* __restrict__ inv,
                                                             const float* __restrict__ resid, const float* __restrict__ scale,
                                                             const float* __restrict__ bias, float* __restrict__ x1,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, void* __restrict__ y, float* __restrict__ mean,
                                                             float* __restrict__ rstd, long long rows, int S, int n_slots,
                                                             int C, OutMap om) {
    constexpr int RPB = THREADS / L;
    const int lane = threadIdx.x % L;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / L;
    const int nchunks = C / 4;
    if (row >= rows) {
        // round 4: y may be written through a token -> slot map (the norm1 + shift + pad + partition of the NEXT block fused into the
        // residual add that ends this one); the slots no token maps to are zero rows, written by the row groups behind the last token
        const long long p = row - rows;
        if (om.map && p < (long long)om.n_pads * om.B) {
            const int b = (int)(p / om.n_pads);
            const size_t yrow = (size_t)b * om.n_out + om.pads[p - (long long)b * om.n_pads];
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int ch = lane + k * L;
                if (ch < nchunks) store4<YDT>(y, yrow * C + 4 * (size_t)ch, f32x4{0.f, 0.f, 0.f, 0.f});
            }
        }
        return;
    }
    const int b = (int)(row / S);
    const int t = (int)(row - (long long)b * S);
    const size_t yrow = om.map ? (size_t)b * om.n_out + om.map[t] : (size_t)row;
    const size_t wrow = ((size_t)b * n_slots + (inv ? inv[t] : t)) * C;
    const float sc = scale ? scale[b] : 1.0f;
    f32x4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = lane + k * L;
        v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ch < nchunks) {
            f32x4 val = load4<WDT>(win, wrow + 4 * (size_t)ch);
            if (bias) val = val + *reinterpret_cast<const f32x4*>(bias + 4 * (size_t)ch);
            if (scale) val = val * sc;
            val = val + *reinterpret_cast<const f32x4*>(resid + (size_t)row * C + 4 * (size_t)ch);
            *reinterpret_cast<f32x4*>(x1 + (size_t)row * C + 4 * (size_t)ch) = val;
            v[k] = val;
            s += (val[0] + val[1]) + (val[2] + val[3]);
        }
    }
    const float mu = row_sum<L>(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = lane + k * L;
        if (ch < nchunks) {
            const f32x4 d = v[k] - mu;
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
    const float rs_ = rsqrtf(row_sum<L>(q) / (float)C + eps);
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = lane + k * L;
        if (ch < nchunks) {
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + 4 * ch);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(beta + 4 * ch);
            store4<YDT>(y, yrow * C + 4 * (size_t)ch, (v[k] - mu) * rs_ * g4 + b4);
        }
    }
    if (lane == 0) {
        mean[row] = mu;
        rstd[row] = rs_;
    }
}

// Backward.  MODE 0 walks source tokens (b, t): dy row = dy[b][inv ? inv[t] : t].  MODE 1 walks merged rows and
// scatters the 4 quarters of dx back to their tokens.  dgamma / dbeta: per-block partial sums, fixed order.
// RSUM: also accumulate sum_rows res_scale[b] * dres[row] (the bias gradient of the Linear whose output, plus bias, was
// added onto the residual stream through the shortcut this LayerNorm's input came along): third partial segment.
template <int MODE, int DYDT, int XDT, int L, int NCH, bool RSUM, bool EX = false>
__global__ __launch_bounds__(THREADS, (EX && NCH == 3) ? 4 : 1) void ln_bwd_kernel(const void* __restrict__ dy, const int32_t* __restrict__ inv,
                                                         const void* __restrict__ x, RowSrc rs,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, void* __restrict__ dx,
                                                         const float* __restrict__ dres,
                                                         const float* __restrict__ res_scale, float* __restrict__ part,
                                                         long long rows, int C, BwdExtra ex = BwdExtra{}) {
    constexpr int RPB = THREADS / L;
    constexpr int NSEG = RSUM ? 3 : 2;
    __shared__ float red[2][THREADS * 4];      // [row group][lane][4 elements] of one chunk column at a time
    const int lane = threadIdx.x % L;
    const int rsub = threadIdx.x / L;
    const int nchunks = C / 4;
    const int n_in = (MODE == 0) ? rs.S : rs.n_out;        // rows walked per image
    f32x4 dg[NCH], db[NCH], dr[RSUM ? NCH : 1];
#pragma unroll
    for (int k = 0; k < NCH; ++k) dg[k] = db[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < (RSUM ? NCH : 1); ++k) dr[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long long row = (long long)blockIdx.x * RPB + rsub; row < rows; row += (long long)gridDim.x * RPB) {
        const int b = (int)(row / n_in);
        const int r = (int)(row - (long long)b * n_in);
        size_t dy_row;
        if constexpr (MODE == 0) dy_row = (size_t)b * rs.n_out + (inv ? inv[r] : r);
        else dy_row = (size_t)row;
        const size_t img_in = (size_t)b * rs.S * (MODE == 0 ? C : C / 4);
        const float mu = mean[row], rs_ = rstd[row];
        const float rsc = (RSUM && res_scale) ? res_scale[b] : 1.0f;
        [[maybe_unused]] unsigned ex_row = 0;                    // element offset of the extra output's row (the launcher checks < 2^31 elements)
        [[maybe_unused]] float ex_sc = 1.0f;
        if constexpr (EX) {
            ex_row = ((unsigned)b * (unsigned)ex.n_rows + (unsigned)(ex.map ? ex.map[r] : r)) * (unsigned)C;
            ex_sc = ex.scale ? ex.scale[b] : 1.0f;
        }
        f32x4 xh[NCH], g[NCH], rv[NCH];
        long long offs[NCH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + k * L;
            xh[k] = g[k] = rv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            offs[k] = -1;
            if (ch < nchunks) {
                offs[k] = src_offset<MODE>(rs, r, ch, C, r);
                f32x4 xv = {0.f, 0.f, 0.f, 0.f};
                if (offs[k] >= 0) {
                    xv = load4<XDT>(x, img_in + (size_t)offs[k]);
                    // the shortcut gradient is requested together with the other two streams (it is only needed
                    // after the row reductions; loading it there exposed its latency)
                    if (dres) rv[k] = *reinterpret_cast<const f32x4*>(dres + img_in + (size_t)offs[k]);
                }
                const f32x4 dyv = load4<DYDT>(dy, dy_row * C + 4 * (size_t)ch);
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + 4 * ch);
                xh[k] = (xv - mu) * rs_;
                g[k] = dyv * g4;
                dg[k] = dg[k] + dyv * xh[k];
                db[k] = db[k] + dyv;
                s1 += (g[k][0] + g[k][1]) + (g[k][2] + g[k][3]);
                s2 += (g[k][0] * xh[k][0] + g[k][1] * xh[k][1]) + (g[k][2] * xh[k][2] + g[k][3] * xh[k][3]);
            }
        }
        s1 = row_sum<L>(s1) / (float)C;
        s2 = row_sum<L>(s2) / (float)C;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + k * L;
            if (ch < nchunks && offs[k] >= 0) {
                f32x4 v = (g[k] - s1 - xh[k] * s2) * rs_;
                // the gradient that reaches x along the residual shortcut, added here instead of in a separate pass
                if (dres) {
                    v = v + rv[k];
                    if constexpr (RSUM) dr[k] = dr[k] + rv[k] * rsc;
                }
                store4<XDT>(dx, img_in + (size_t)offs[k], v);
                if constexpr (EX) store4<PSWIN_BF16>(ex.ex, ex_row + 4u * (unsigned)ch, v * ex_sc);
            }
        }
    }
    if constexpr (EX) {             // the slots no token maps to: zero rows (what the separate window gather writes there)
        const long long npad = (long long)ex.n_pads * ex.B;
        for (long long p = (long long)blockIdx.x * RPB + rsub; p < npad; p += (long long)gridDim.x * RPB) {
            const int b = (int)(p / ex.n_pads);
            const size_t er = (size_t)b * ex.n_rows + ex.pads[p - (long long)b * ex.n_pads];
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int ch = lane + k * L;
                if (ch < nchunks) store4<PSWIN_BF16>(ex.ex, er * C + 4 * (size_t)ch, f32x4{0.f, 0.f, 0.f, 0.f});
            }
        }
    }
    // block reduction of dgamma / dbeta over the RPB row groups (fixed order), then one partial row per block
    float* outp = part + (size_t)blockIdx.x * NSEG * C;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = lane + k * L;
        __syncthreads();
        if (ch < nchunks) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                red[0][(rsub * L + lane) * 4 + e] = dg[k][e];
                red[1][(rsub * L + lane) * 4 + e] = db[k][e];
            }
        }
        __syncthreads();
        if (rsub == 0 && ch < nchunks) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = 0.f, c = 0.f;
#pragma unroll 4
                for (int q = 0; q < RPB; ++q) {
                    a += red[0][(q * L + lane) * 4 + e];
                    c += red[1][(q * L + lane) * 4 + e];
                }
                outp[4 * ch + e] = a;
                outp[C + 4 * ch + e] = c;
            }
        }
        if constexpr (RSUM) {
            __syncthreads();
            if (ch < nchunks) {
#pragma unroll
                for (int e = 0; e < 4; ++e) red[0][(rsub * L + lane) * 4 + e] = dr[k][e];
            }
            __syncthreads();
            if (rsub == 0 && ch < nchunks) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = 0.f;
#pragma unroll 4
                    for (int q = 0; q < RPB; ++q) a += red[0][(q * L + lane) * 4 + e];
                    outp[2 * C + 4 * ch + e] = a;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Output norms (HOT:975-977: norm{i}(x_out) -> view(B, H, W, C) -> permute(0, 3, 1, 2).contiguous()):
// LayerNorm written directly in NCHW.  A block normalises RPB = THREADS / L consecutive tokens, parks the result in an
// LDS tile and writes it out channel-major, RPB * 4 bytes contiguous per channel, instead of a token-major store plus a
// separate strided transpose pass over the (largest) output tensor; the backward kernel reads the NCHW gradient the
// same way.  fp32 in, fp32 out; S % RPB == 0 (the host falls back to LN + copy otherwise).
// ---------------------------------------------------------------------------------------------
template <int L, int NCH>
__global__ __launch_bounds__(THREADS) void ln_nchw_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps,
                                                              float* __restrict__ y, float* __restrict__ mean,
                                                              float* __restrict__ rstd, int S, int C,
                                                              const unsigned short* __restrict__ win, const float* __restrict__ scale,
                                                              const float* __restrict__ bias, float* __restrict__ x1) {
    constexpr int RPB = THREADS / L;
    extern __shared__ float tile[];                 // [C][RPB + 1]
    const int lane = threadIdx.x % L, rsub = threadIdx.x / L;
    const long long row = (long long)blockIdx.x * RPB + rsub;       // grid covers B * S exactly
    const int nchunks = C / 4;
    // win != nullptr (round 4): the rows are x1 = x + scale_b * (win + bias) -- the residual add that closes a stage (x: the shortcut,
    // win: the bf16 MLP branch in token order) -- written to x1 and normalised in the same pass; the arithmetic and its order are those
    // of window_scatter_add_kernel (as in ln_add_fwd_kernel): bitwise the same x1
    const float sc = (win && scale) ? scale[(long long)blockIdx.x * RPB / S] : 1.0f;
    f32x4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = lane + k * L;
        v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ch < nchunks) {
            if (win) {
                f32x4 val = load4<PSWIN_BF16>(win, (size_t)row * C + 4 * (size_t)ch);
                if (bias) val = val + *reinterpret_cast<const f32x4*>(bias + 4 * (size_t)ch);
                if (scale) val = val * sc;
                val = val + *reinterpret_cast<const f32x4*>(x + (size_t)row * C + 4 * ch);
                *reinterpret_cast<f32x4*>(x1 + (size_t)row * C + 4 * (size_t)ch) = val;
                v[k] = val;
            } else {
                v[k] = *reinterpret_cast<const f32x4*>(x + (size_t)row * C + 4 * ch);
            }
            s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
        }
    }
    const float mu = row_sum<L>(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = lane + k * L;
        if (ch < nchunks) {
            const f32x4 d = v[k] - mu;
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
    const float rs_ = rsqrtf(row_sum<L>(q) / (float)C + eps);
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = lane + k * L;
        if (ch < nchunks) {
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + 4 * ch);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(beta + 4 * ch);
            const f32x4 o = (v[k] - mu) * rs_ * g4 + b4;
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[(4 * ch + e) * (RPB + 1) + rsub] = o[e];
        }
    }
    if (lane == 0) {
        mean[row] = mu;
        rstd[row] = rs_;
    }
    __syncthreads();
    const long long row0 = (long long)blockIdx.x * RPB;
    const int b = (int)(row0 / S);
    const int t0 = (int)(row0 - (long long)b * S);
    float* yb = y + (size_t)b * C * S + t0;
    constexpr int Q = RPB / 4;                       // 16-byte groups per channel
    for (int i = threadIdx.x; i < C * Q; i += THREADS) {
        const int c = i / Q, r4 = i - c * Q;
        const float* tp = tile + c * (RPB + 1) + 4 * r4;
        *reinterpret_cast<f32x4*>(yb + (size_t)c * S + 4 * r4) = f32x4{tp[0], tp[1], tp[2], tp[3]};
    }
}

template <int L, int NCH>
__global__ __launch_bounds__(THREADS) void ln_nchw_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ dres,
                                                              float* __restrict__ dx, float* __restrict__ part,
                                                              long long rows, int S, int C, void* __restrict__ ex,
                                                              const float* __restrict__ ex_scale) {
    constexpr int RPB = THREADS / L;
    constexpr int Q = RPB / 4;
    extern __shared__ float tile[];                 // [RPB][C + 4] then the partial-sum staging
    __shared__ float red[2][THREADS * 4];
    const int lane = threadIdx.x % L, rsub = threadIdx.x / L;
    const int nchunks = C / 4, LD = C + 4;
    f32x4 dg[NCH], db[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) dg[k] = db[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long long row0 = (long long)blockIdx.x * RPB; row0 < rows; row0 += (long long)gridDim.x * RPB) {
        const int b = (int)(row0 / S);
        const int t0 = (int)(row0 - (long long)b * S);
        const float* dyb = dy + (size_t)b * C * S + t0;
        __syncthreads();
        for (int i = threadIdx.x; i < C * Q; i += THREADS) {
            const int c = i / Q, r4 = i - c * Q;
            const f32x4 t = *reinterpret_cast<const f32x4*>(dyb + (size_t)c * S + 4 * r4);
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[(4 * r4 + e) * LD + c] = t[e];
        }
        __syncthreads();
        const long long row = row0 + rsub;
        const float mu = mean[row], rs_ = rstd[row];
        f32x4 xh[NCH], g[NCH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + k * L;
            xh[k] = g[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ch < nchunks) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)row * C + 4 * ch);
                const f32x4 dyv = *reinterpret_cast<const f32x4*>(tile + rsub * LD + 4 * ch);
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + 4 * ch);
                xh[k] = (xv - mu) * rs_;
                g[k] = dyv * g4;
                dg[k] = dg[k] + dyv * xh[k];
                db[k] = db[k] + dyv;
                s1 += (g[k][0] + g[k][1]) + (g[k][2] + g[k][3]);
                s2 += (g[k][0] * xh[k][0] + g[k][1] * xh[k][1]) + (g[k][2] * xh[k][2] + g[k][3] * xh[k][3]);
            }
        }
        s1 = row_sum<L>(s1) / (float)C;
        s2 = row_sum<L>(s2) / (float)C;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + k * L;
            if (ch < nchunks) {
                f32x4 v = (g[k] - s1 - xh[k] * s2) * rs_;
                if (dres) v = v + *reinterpret_cast<const f32x4*>(dres + (size_t)row * C + 4 * ch);
                *reinterpret_cast<f32x4*>(dx + (size_t)row * C + 4 * ch) = v;
                // ex (round 4): bf16(scale[b] * dx) in token order -- the gradient of the stage's closing MLP branch (the backward of
                // window_scatter_add's cast), written while dx is in registers instead of by a pswin_window_gather pass
                if (ex) store4<PSWIN_BF16>(ex, (size_t)row * C + 4 * (size_t)ch, ex_scale ? v * ex_scale[b] : v);
            }
        }
    }
    float* outp = part + (size_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = lane + k * L;
        __syncthreads();
        if (ch < nchunks) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                red[0][(rsub * L + lane) * 4 + e] = dg[k][e];
                red[1][(rsub * L + lane) * 4 + e] = db[k][e];
            }
        }
        __syncthreads();
        if (rsub == 0 && ch < nchunks) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = 0.f, c = 0.f;
#pragma unroll 4
                for (int q = 0; q < RPB; ++q) {
                    a += red[0][(q * L + lane) * 4 + e];
                    c += red[1][(q * L + lane) * 4 + e];
                }
                outp[4 * ch + e] = a;
                outp[C + 4 * ch + e] = c;
            }
        }
    }
}

// out_k[c] = sum_r part[r][k * C + c] for the nseg (2 or 3) segments of rows of nseg * C floats; 16 columns x 64 row lanes
__global__ void colsum_seg_kernel(const float* __restrict__ part, int R, int C, int nseg, float* __restrict__ out_a,
                                  float* __restrict__ out_b, float* __restrict__ out_c) {
    __shared__ float red[64][16];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const int N = nseg * C;
    float s0 = 0.f, s1 = 0.f;
    if (c < N) {
        int r = rl;
        for (; r + 64 < R; r += 128) {
            s0 += part[(size_t)r * N + c];
            s1 += part[(size_t)(r + 64) * N + c];
        }
        for (; r < R; r += 64) s0 += part[(size_t)r * N + c];
    }
    red[rl][cl] = s0 + s1;
    __syncthreads();
    if (rl < 16) red[rl][cl] = (red[rl][cl] + red[rl + 16][cl]) + (red[rl + 32][cl] + red[rl + 48][cl]);
    __syncthreads();
    if (rl == 0 && c < N) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][cl];
        if (c < C) out_a[c] = s;
        else if (c < 2 * C) out_b[c - C] = s;
        else out_c[c - 2 * C] = s;
    }
}

inline void launch_colsum_seg(const float* part, int R, int C, int nseg, float* out_a, float* out_b, float* out_c,
                              hipStream_t st) {
    hipLaunchKernelGGL(colsum_seg_kernel, dim3((nseg * C + 15) / 16), dim3(1024), 0, st, part, R, C, nseg, out_a, out_b, out_c);
}

// up to 4 chunks of 4 elements per lane (8 for rows wider than 1024 elements: PatchMerging of C >= 384)
inline int pick_lanes(int C) {
    int L = 2;
    while (L * 16 < C && L < 64) L *= 2;
    return L;
}
inline bool wide_row(int C) { return C > 1024; }
constexpr int MAX_C = 2048;

// Persistent grid of the backward kernels: exactly the resident capacity (256 CUs x 4 blocks of 4 waves at ~125 VGPRs).
// 1536 blocks ran as 1.5 rounds (the last half round leaves half of the chip idle: -1.2 % end to end), 512 leave two
// of the four wave slots per SIMD empty (3.4 TB/s).  The partial rows are summed by the grouped end-of-pass reduction.
constexpr int BWD_MAX_BLOCKS = 1024;

inline int bwd_blocks(long long rows, int L) {
    const int rpb = THREADS / L;
    long long nb = (rows + rpb - 1) / rpb;
    return (int)(nb < BWD_MAX_BLOCKS ? nb : BWD_MAX_BLOCKS);
}

template <int MODE, int XDT, int YDT>
int launch_fwd(int L, const void* x, const RowSrc& rs, const float* gamma, const float* beta, float eps, void* y,
               float* mean, float* rstd, long long rows, int C, hipStream_t st, const float* add = nullptr) {
#define PSWIN_LN_FWD(LL)                                                                                          \
    case LL: {                                                                                                    \
        const int rpb = THREADS / LL;                                                                             \
        hipLaunchKernelGGL((ln_fwd_kernel<MODE, XDT, YDT, LL, 4>), dim3((unsigned)((rows + rpb - 1) / rpb)),      \
                           dim3(THREADS), 0, st, x, rs, gamma, beta, eps, y, mean, rstd, rows, C, add);           \
        break;                                                                                                    \
    }
    if (wide_row(C)) {
        hipLaunchKernelGGL((ln_fwd_kernel<MODE, XDT, YDT, 64, 8>), dim3((unsigned)((rows + 3) / 4)), dim3(THREADS), 0,
                           st, x, rs, gamma, beta, eps, y, mean, rstd, rows, C, add);
        PSWIN_LAUNCH_RET();
    }
    switch (L) {
        PSWIN_LN_FWD(2) PSWIN_LN_FWD(4) PSWIN_LN_FWD(8) PSWIN_LN_FWD(16) PSWIN_LN_FWD(32) PSWIN_LN_FWD(64)
        default: return PSWIN_ERR_ARG;
    }
#undef PSWIN_LN_FWD
    PSWIN_LAUNCH_RET();
}

template <int MODE, int DYDT, int XDT, bool RSUM>
int launch_bwd(int L, const void* dy, const int32_t* inv, const void* x, const RowSrc& rs, const float* mean,
               const float* rstd, const float* gamma, void* dx, const float* dres, const float* res_scale, float* part,
               long long rows, int C, int blocks, hipStream_t st, BwdExtra ex = BwdExtra{}) {
    constexpr bool CAN_EX = MODE == 0 && XDT == PSWIN_F32;
#define PSWIN_LN_BWD(LL)                                                                                                \
    case LL:                                                                                                            \
        if constexpr (CAN_EX) {                                                                                         \
            if (ex.ex) {                                                                                                \
                if (C / 4 <= 3 * LL)                                                                                    \
                    hipLaunchKernelGGL((ln_bwd_kernel<MODE, DYDT, XDT, LL, 3, RSUM, true>), dim3(blocks), dim3(THREADS), 0, st, dy, inv, \
                                       x, rs, mean, rstd, gamma, dx, dres, res_scale, part, rows, C, ex);               \
                else                                                                                                    \
                    hipLaunchKernelGGL((ln_bwd_kernel<MODE, DYDT, XDT, LL, 4, RSUM, true>), dim3(blocks), dim3(THREADS), 0, st, dy, inv, \
                                       x, rs, mean, rstd, gamma, dx, dres, res_scale, part, rows, C, ex);               \
                break;                                                                                                  \
            }                                                                                                           \
        }                                                                                                               \
        if (C / 4 <= 3 * LL)      /* 3 chunks per lane (C = 96, 192, 384, 768): 20 registers less, 4 waves per SIMD */      \
            hipLaunchKernelGGL((ln_bwd_kernel<MODE, DYDT, XDT, LL, 3, RSUM>), dim3(blocks), dim3(THREADS), 0, st, dy, inv, \
                               x, rs, mean, rstd, gamma, dx, dres, res_scale, part, rows, C, BwdExtra{});               \
        else                                                                                                            \
            hipLaunchKernelGGL((ln_bwd_kernel<MODE, DYDT, XDT, LL, 4, RSUM>), dim3(blocks), dim3(THREADS), 0, st, dy, inv, \
                               x, rs, mean, rstd, gamma, dx, dres, res_scale, part, rows, C, BwdExtra{});               \
        break;
    if (wide_row(C)) {
        if (ex.ex) return PSWIN_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((ln_bwd_kernel<MODE, DYDT, XDT, 64, 8, RSUM>), dim3(blocks), dim3(THREADS), 0, st, dy, inv, x, rs,
                           mean, rstd, gamma, dx, dres, res_scale, part, rows, C, BwdExtra{});
        PSWIN_LAUNCH_RET();
    }
    if (ex.ex && !CAN_EX) return PSWIN_ERR_UNSUPPORTED;
    switch (L) {
        PSWIN_LN_BWD(2) PSWIN_LN_BWD(4) PSWIN_LN_BWD(8) PSWIN_LN_BWD(16) PSWIN_LN_BWD(32) PSWIN_LN_BWD(64)
        default: return PSWIN_ERR_ARG;
    }
#undef PSWIN_LN_BWD
    PSWIN_LAUNCH_RET();
}

template <typename F>
inline int dispatch2(int a, int b, F&& f) {
    if (a == PSWIN_F32 && b == PSWIN_F32) return f(std::integral_constant<int, PSWIN_F32>(), std::integral_constant<int, PSWIN_F32>());
    if (a == PSWIN_F32 && b == PSWIN_BF16) return f(std::integral_constant<int, PSWIN_F32>(), std::integral_constant<int, PSWIN_BF16>());
    if (a == PSWIN_BF16 && b == PSWIN_F32) return f(std::integral_constant<int, PSWIN_BF16>(), std::integral_constant<int, PSWIN_F32>());
    return f(std::integral_constant<int, PSWIN_BF16>(), std::integral_constant<int, PSWIN_BF16>());
}

}  // namespace

extern "C" int pswin_ln_workspace(long long rows, int C) {
    if (rows <= 0 || C <= 0 || C % 8) return PSWIN_ERR_ARG;
    return bwd_blocks(rows, pick_lanes(C)) * 3 * C;
}

extern "C" int pswin_ln_partial_rows(long long rows, int C) {
    if (rows <= 0 || C <= 0 || C % 8) return PSWIN_ERR_ARG;
    return bwd_blocks(rows, pick_lanes(C));
}

extern "C" int pswin_ln_gather_fwd_add(const void* x, int x_dtype, const int32_t* map, const float* gamma,
                                       const float* beta, float eps, const float* add_rows, void* y, int y_dtype, float* mean,
                                       float* rstd, int B, int S, int n_out, int C, void* stream) {
    PSWIN_CHECK_ARG(x && gamma && beta && y && mean && rstd && B > 0 && S > 0 && n_out > 0);
    PSWIN_CHECK_ARG(valid_dtype(x_dtype) && valid_dtype(y_dtype));
    PSWIN_CHECK_ARG(C >= 8 && C % 8 == 0 && C <= MAX_C && aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta));
    PSWIN_CHECK_ARG(map || n_out == S);
    PSWIN_CHECK_ARG(!add_rows || (!map && aligned16(add_rows)));      // the added rows follow the token order of an image
    RowSrc rs = {0, map, S, n_out, 0, 0, 0};
    const long long rows = (long long)B * n_out;
    const int L = pick_lanes(C);
    return dispatch2(x_dtype, y_dtype, [&](auto xd, auto yd) {
        return launch_fwd<0, decltype(xd)::value, decltype(yd)::value>(L, x, rs, gamma, beta, eps, y, mean, rstd, rows,
                                                                       C, (hipStream_t)stream, add_rows);
    });
}

extern "C" int pswin_ln_gather_fwd(const void* x, int x_dtype, const int32_t* map, const float* gamma,
                                   const float* beta, float eps, void* y, int y_dtype, float* mean, float* rstd, int B,
                                   int S, int n_out, int C, void* stream) {
    return pswin_ln_gather_fwd_add(x, x_dtype, map, gamma, beta, eps, nullptr, y, y_dtype, mean, rstd, B, S, n_out, C, stream);
}

extern "C" int pswin_scatter_add_ln_fwd_map(const void* win, int win_dtype, const int32_t* inv, const float* resid,
                                            const float* scale, const float* bias, float* x1, const float* gamma,
                                            const float* beta, float eps, void* y, int y_dtype, float* mean, float* rstd, int B,
                                            int S, int n_slots, int C, const int32_t* out_map, int n_out, const int32_t* out_pads,
                                            int n_out_pads, void* stream) {
    PSWIN_CHECK_ARG(win && resid && x1 && gamma && beta && y && mean && rstd && B > 0 && S > 0 && n_slots > 0);
    PSWIN_CHECK_ARG(valid_dtype(win_dtype) && valid_dtype(y_dtype) && (inv || n_slots == S));
    PSWIN_CHECK_ARG(C >= 8 && C % 8 == 0 && C <= 1024 && aligned16(win) && aligned16(resid) && aligned16(x1) && aligned16(y));
    PSWIN_CHECK_ARG(aligned16(gamma) && aligned16(beta) && aligned16(bias));
    PSWIN_CHECK_ARG(out_map ? (n_out >= S && n_out_pads == n_out - S && (out_pads || n_out_pads == 0)) : (n_out == S && n_out_pads == 0));
    const OutMap om = {out_map, out_pads, n_out, n_out_pads, B};
    const long long rows = (long long)B * S;
    const long long groups = rows + (out_map ? (long long)B * n_out_pads : 0);     // row groups: tokens, then the zero slots
    const int L = pick_lanes(C);
    hipStream_t st = (hipStream_t)stream;
    return dispatch2(win_dtype, y_dtype, [&](auto wd, auto yd) {
#define PSWIN_LN_ADD(LL)                                                                                                   \
    case LL: {                                                                                                             \
        const int rpb = THREADS / LL;                                                                                      \
        hipLaunchKernelGGL((ln_add_fwd_kernel<decltype(wd)::value, decltype(yd)::value, LL, 4>),                            \
                           dim3((unsigned)((groups + rpb - 1) / rpb)), dim3(THREADS), 0, st, win, inv, resid, scale, bias, x1, \
                           gamma, beta, eps, y, mean, rstd, rows, S, n_slots, C, om);                                      \
        break;                                                                                                             \
    }
        switch (L) {
            PSWIN_LN_ADD(2) PSWIN_LN_ADD(4) PSWIN_LN_ADD(8) PSWIN_LN_ADD(16) PSWIN_LN_ADD(32) PSWIN_LN_ADD(64)
            default: return (int)PSWIN_ERR_ARG;
        }
#undef PSWIN_LN_ADD
        hipError_t e__ = hipGetLastError();
        return e__ == hipSuccess ? (int)PSWIN_OK : (int)e__;
    });
}

extern "C" int pswin_scatter_add_ln_fwd(const void* win, int win_dtype, const int32_t* inv, const float* resid,
                                        const float* scale, const float* bias, float* x1, const float* gamma,
                                        const float* beta, float eps, void* y, int y_dtype, float* mean, float* rstd, int B,
                                        int S, int n_slots, int C, void* stream) {
    return pswin_scatter_add_ln_fwd_map(win, win_dtype, inv, resid, scale, bias, x1, gamma, beta, eps, y, y_dtype, mean, rstd, B, S, n_slots, C,
                                        nullptr, S, nullptr, 0, stream);
}

extern "C" int pswin_ln_gather_bwd_ex(const void* dy, int dy_dtype, const int32_t* inv, const void* x, int x_dtype,
                                      const float* mean, const float* rstd, const float* gamma, const float* dres,
                                      const float* res_scale, float* dres_sum, void* dx, float* dgamma, float* dbeta,
                                      float* workspace, int B, int S, int n_out, int C, void* ex, const int32_t* ex_map, int ex_rows,
                                      const float* ex_scale, const int32_t* ex_pads, int n_ex_pads, void* stream) {
    PSWIN_CHECK_ARG(dy && x && mean && rstd && gamma && dx && workspace && B > 0 && S > 0 && n_out > 0);
    PSWIN_CHECK_ARG((dgamma && dbeta) || (!dgamma && !dbeta));
    PSWIN_CHECK_ARG(valid_dtype(x_dtype) && valid_dtype(dy_dtype));
    PSWIN_CHECK_ARG(C >= 8 && C % 8 == 0 && C <= MAX_C && aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(gamma));
    PSWIN_CHECK_ARG(inv || n_out == S);
    PSWIN_CHECK_ARG(!dres || (x_dtype == PSWIN_F32 && aligned16(dres)));
    PSWIN_CHECK_ARG(!dres_sum || dres);
    if (ex) {
        PSWIN_CHECK_ARG(x_dtype == PSWIN_F32 && aligned16(ex) && !wide_row(C) && (long long)B * ex_rows * C < 0x7fffffffll);
        PSWIN_CHECK_ARG(ex_map ? (ex_rows >= S && n_ex_pads == ex_rows - S && (ex_pads || n_ex_pads == 0)) : (ex_rows == S && n_ex_pads == 0));
    }
    const BwdExtra be = {ex, ex_map, ex_scale, ex_pads, ex_rows, ex ? n_ex_pads : 0, B};
    RowSrc rs = {0, nullptr, S, n_out, 0, 0, 0};
    const long long rows = (long long)B * S;
    const int L = pick_lanes(C);
    const int blocks = bwd_blocks(rows, L);
    int rc;
    if (dres_sum) {
        rc = dispatch2(dy_dtype, PSWIN_F32, [&](auto dd, auto xd) {
            return launch_bwd<0, decltype(dd)::value, PSWIN_F32, true>(L, dy, inv, x, rs, mean, rstd, gamma, dx, dres, res_scale,
                                                                       workspace, rows, C, blocks, (hipStream_t)stream, be);
        });
    } else {
        rc = dispatch2(dy_dtype, x_dtype, [&](auto dd, auto xd) {
            return launch_bwd<0, decltype(dd)::value, decltype(xd)::value, false>(L, dy, inv, x, rs, mean, rstd, gamma, dx, dres,
                                                                                  nullptr, workspace, rows, C, blocks,
                                                                                  (hipStream_t)stream, be);
        });
    }
    if (rc) return rc;
    // partial rows are [dgamma(C) | dbeta(C) | dres_sum(C)]; the outputs may live in different buffers.
    // dgamma == dbeta == NULL: the caller sums the partial rows itself (pswin_reduce_jobs); dres_sum then only selects
    // the 3-segment row layout and is not written.
    if (dgamma) launch_colsum_seg(workspace, blocks, C, dres_sum ? 3 : 2, dgamma, dbeta, dres_sum, (hipStream_t)stream);
    PSWIN_LAUNCH_RET();
}

extern "C" int pswin_ln_gather_bwd(const void* dy, int dy_dtype, const int32_t* inv, const void* x, int x_dtype,
                                   const float* mean, const float* rstd, const float* gamma, const float* dres,
                                   const float* res_scale, float* dres_sum, void* dx, float* dgamma, float* dbeta,
                                   float* workspace, int B, int S, int n_out, int C, void* stream) {
    return pswin_ln_gather_bwd_ex(dy, dy_dtype, inv, x, x_dtype, mean, rstd, gamma, dres, res_scale, dres_sum, dx, dgamma, dbeta, workspace, B, S,
                                  n_out, C, nullptr, nullptr, S, nullptr, nullptr, 0, stream);
}

extern "C" int pswin_ln_patch_merge_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, float eps,
                                        void* y, int y_dtype, float* mean, float* rstd, int B, int H, int W, int C,
                                        void* stream) {
    PSWIN_CHECK_ARG(x && gamma && beta && y && mean && rstd && B > 0 && H > 0 && W > 0);
    PSWIN_CHECK_ARG(valid_dtype(x_dtype) && valid_dtype(y_dtype));
    const int C4 = 4 * C;
    PSWIN_CHECK_ARG(C >= 16 && C % 16 == 0 && C4 <= MAX_C && aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta));
    const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
    RowSrc rs = {1, nullptr, H * W, H2 * W2, H, W, W2};
    const long long rows = (long long)B * H2 * W2;
    const int L = pick_lanes(C4);
    return dispatch2(x_dtype, y_dtype, [&](auto xd, auto yd) {
        return launch_fwd<1, decltype(xd)::value, decltype(yd)::value>(L, x, rs, gamma, beta, eps, y, mean, rstd, rows,
                                                                       C4, (hipStream_t)stream);
    });
}

extern "C" int pswin_ln_patch_merge_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean,
                                        const float* rstd, const float* gamma, void* dx, float* dgamma, float* dbeta,
                                        float* workspace, int B, int H, int W, int C, void* stream) {
    PSWIN_CHECK_ARG(dy && x && mean && rstd && gamma && dx && workspace && B > 0 && H > 0 && W > 0);
    PSWIN_CHECK_ARG((dgamma && dbeta) || (!dgamma && !dbeta));
    PSWIN_CHECK_ARG(valid_dtype(x_dtype) && valid_dtype(dy_dtype));
    const int C4 = 4 * C;
    PSWIN_CHECK_ARG(C >= 16 && C % 16 == 0 && C4 <= MAX_C && aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(gamma));
    const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
    RowSrc rs = {1, nullptr, H * W, H2 * W2, H, W, W2};
    const long long rows = (long long)B * H2 * W2;
    const int L = pick_lanes(C4);
    const int blocks = bwd_blocks(rows, L);
    int rc = dispatch2(dy_dtype, x_dtype, [&](auto dd, auto xd) {
        return launch_bwd<1, decltype(dd)::value, decltype(xd)::value, false>(L, dy, nullptr, x, rs, mean, rstd, gamma, dx,
                                                                              nullptr, nullptr, workspace, rows, C4, blocks,
                                                                              (hipStream_t)stream);
    });
    if (rc) return rc;
    if (dgamma) launch_colsum_seg(workspace, blocks, C4, 2, dgamma, dbeta, nullptr, (hipStream_t)stream);
    PSWIN_LAUNCH_RET();
}

extern "C" int pswin_ln_nchw_supported(int S, int C) {
    if (S <= 0 || C < 8 || C % 8 || C > 1024) return 0;
    const int L = pick_lanes(C);
    const int rpb = THREADS / L;
    return (rpb >= 4 && S % rpb == 0) ? 1 : 0;
}

extern "C" int pswin_scatter_add_ln_nchw_fwd(const void* win_bf16, const float* x, const float* scale, const float* bias, float* x1,
                                             const float* gamma, const float* beta, float eps, float* y, float* mean, float* rstd,
                                             int B, int S, int C, void* stream) {
    const void* win = win_bf16;
    PSWIN_CHECK_ARG(x && gamma && beta && y && mean && rstd && B > 0 && pswin_ln_nchw_supported(S, C));
    PSWIN_CHECK_ARG(aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta));
    PSWIN_CHECK_ARG(win ? (x1 && aligned16(win) && aligned16(x1) && aligned16(bias)) : (!x1 && !scale && !bias));
    const int L = pick_lanes(C), rpb = THREADS / L;
    const long long rows = (long long)B * S;
    const unsigned grid = (unsigned)(rows / rpb);
    const size_t lds = (size_t)C * (rpb + 1) * sizeof(float);
#define PSWIN_LN_NCHW_F(LL)                                                                                               \
    case LL:                                                                                                              \
        hipLaunchKernelGGL((ln_nchw_fwd_kernel<LL, 4>), dim3(grid), dim3(THREADS), lds, (hipStream_t)stream, x, gamma, beta, \
                           eps, y, mean, rstd, S, C, reinterpret_cast<const unsigned short*>(win), scale, bias, x1);       \
        break;
    switch (L) {
        PSWIN_LN_NCHW_F(2) PSWIN_LN_NCHW_F(4) PSWIN_LN_NCHW_F(8) PSWIN_LN_NCHW_F(16) PSWIN_LN_NCHW_F(32) PSWIN_LN_NCHW_F(64)
        default: return PSWIN_ERR_ARG;
    }
#undef PSWIN_LN_NCHW_F
    PSWIN_LAUNCH_RET();
}

extern "C" int pswin_ln_nchw_fwd(const float* x, const float* gamma, const float* beta, float eps, float* y, float* mean,
                                 float* rstd, int B, int S, int C, void* stream) {
    return pswin_scatter_add_ln_nchw_fwd(nullptr, x, nullptr, nullptr, nullptr, gamma, beta, eps, y, mean, rstd, B, S, C, stream);
}

extern "C" int pswin_ln_nchw_bwd_ex(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                    const float* dres, float* dx, float* dgamma, float* dbeta, float* workspace, void* ex_bf16,
                                    const float* ex_scale, int B, int S, int C, void* stream) {
    PSWIN_CHECK_ARG(dy && x && mean && rstd && gamma && dx && workspace && B > 0 && pswin_ln_nchw_supported(S, C));
    PSWIN_CHECK_ARG((dgamma && dbeta) || (!dgamma && !dbeta));
    PSWIN_CHECK_ARG(aligned16(dy) && aligned16(x) && aligned16(dx) && aligned16(gamma) && aligned16(dres) && aligned16(ex_bf16));
    const int L = pick_lanes(C), rpb = THREADS / L;
    const long long rows = (long long)B * S;
    const int blocks = bwd_blocks(rows, L);
    const size_t lds = (size_t)rpb * (C + 4) * sizeof(float);
#define PSWIN_LN_NCHW_B(LL)                                                                                              \
    case LL:                                                                                                             \
        hipLaunchKernelGGL((ln_nchw_bwd_kernel<LL, 4>), dim3(blocks), dim3(THREADS), lds, (hipStream_t)stream, dy, x, mean, \
                           rstd, gamma, dres, dx, workspace, rows, S, C, ex_bf16, ex_scale);                             \
        break;
    switch (L) {
        PSWIN_LN_NCHW_B(2) PSWIN_LN_NCHW_B(4) PSWIN_LN_NCHW_B(8) PSWIN_LN_NCHW_B(16) PSWIN_LN_NCHW_B(32) PSWIN_LN_NCHW_B(64)
        default: return PSWIN_ERR_ARG;
    }
#undef PSWIN_LN_NCHW_B
    if (dgamma) launch_colsum_seg(workspace, blocks, C, 2, dgamma, dbeta, nullptr, (hipStream_t)stream);
    PSWIN_LAUNCH_RET();
}

extern "C" int pswin_ln_nchw_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                 const float* dres, float* dx, float* dgamma, float* dbeta, float* workspace, int B, int S,
                                 int C, void* stream) {
    return pswin_ln_nchw_bwd_ex(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, workspace, nullptr, nullptr, B, S, C, stream);
}
