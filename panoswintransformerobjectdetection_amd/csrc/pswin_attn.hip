// 7x7-window multi-head attention for gfx950 (MI355X): forward, backward and the PanoSwin score bias.
//
// Replaces BasicWindowAttention.forward between the qkv and proj Linear layers (HOT:288-308:
// q*scale, q@k^T, + great-circle/relative-position bias, + shifted-window mask, softmax, @v) and
// PitchAttentionModule._attention (HOT:1226-1234).  The reference materialises five
// [B*nW, heads, 49, 49] fp32 tensors per block for this; here one wave owns one (window, head) pair end
// to end: scores, softmax and the P.V product never leave registers.
// HOT = mmdet/models/backbones/simple_panoswin_transformer.py of the reference.
//
// Design (CDNA4, wave64):
//   * 49 tokens are padded to 64 = 4 MFMA tiles of 16.  One wave per (bias window wb, head h) work item;
//     the wave loops over the images of the batch that share that bias tile (window n = rep*nb + wb), so the
//     bias is read once per work item and stays in 64 VGPRs as the MFMA C operand: S + bias costs nothing.
//   * the bias tile carries -inf in the padded key columns, so padding needs no masking code.
//   * forward computes S^T = K.Q^T (key on the accumulator row, query on the lane): a score row lives in one
//     lane (+ its 3 partner lanes), softmax needs 2 cross-lane steps, and P is already the B operand of
//     O^T = V^T.P^T.  V^T comes from LDS through ds_read_b64_tr_b16 (bf16) / scalar LDS reads (f32).
//   * backward computes S = Q.K^T and dP = dO.V^T with the query on the accumulator row: P and dS are then
//     already the B operands of dV^T = dO^T.P and dK^T = Q^T.dS; only dS crosses LDS once (for dQ).
//     dBias = sum over the batch loop of dS stays in 64 VGPRs and is written once per work item.
//   * dtype f32 uses the exact-f32 MFMA (v_mfma_f32_16x16x4_f32, k-ordered fmaf chain) so the f32 path
//     matches the PyTorch reference to summation order; dtype bf16 uses v_mfma_f32_16x16x32_bf16 with f32
//     softmax / accumulation.
//   * Q/K/V/dO fragments are loaded straight from HBM in MFMA operand layout (lane = row & 15, 8 contiguous
//     head-dim elements per lane = one 16-byte load for bf16): no staging pass for the row-wise operands.
#include "pswin_common.hpp"

using namespace pswin;

namespace {

constexpr int TOK = PSWIN_WTOK;    // 49
constexpr int PADT = PSWIN_WPAD;   // 64
constexpr int HD = PSWIN_HEAD_DIM; // 32
constexpr float LOG2E = 1.4426950408889634f;

// ---------------------------------------------------------------------------------------------
// fragments: 8 head-dim (or key) elements of one row, the unit both MFMA flavours contract over
// ---------------------------------------------------------------------------------------------
template <int DT>
struct Frag;
template <>
struct Frag<PSWIN_BF16> {
    bf16x8 v;
};
template <>
struct Frag<PSWIN_F32> {
    float v[8];
};

template <int DT>
__device__ inline Frag<DT> zero_frag() {
    Frag<DT> f;
    if constexpr (DT == PSWIN_BF16) {
        u32x4 z = {0u, 0u, 0u, 0u};
        f.v = __builtin_bit_cast(bf16x8, z);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) f.v[i] = 0.f;
    }
    return f;
}

// 8 consecutive elements starting at element offset `off` (16-byte aligned)
template <int DT>
__device__ inline Frag<DT> load_frag(const void* base, size_t off, bool valid) {
    Frag<DT> f = zero_frag<DT>();
    if (valid) {
        if constexpr (DT == PSWIN_BF16) {
            u32x4 raw = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(base) + off);
            f.v = __builtin_bit_cast(bf16x8, raw);
        } else {
            const float* p = reinterpret_cast<const float*>(base) + off;
            f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f.v[i] = a[i];
                f.v[4 + i] = b[i];
            }
        }
    }
    return f;
}

template <int DT>
__device__ inline Frag<DT> scale_frag(Frag<DT> f, float s) {
    if constexpr (DT == PSWIN_BF16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f.v[i] = (__bf16)((float)f.v[i] * s);   // bf16 q*scale, as torch would round it
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) f.v[i] *= s;
    }
    return f;
}

// two accumulator quads -> the 8-element operand of the next MFMA (k order: lo[0..3], hi[0..3])
template <int DT>
__device__ inline Frag<DT> pack_frag(f32x4 lo, f32x4 hi) {
    Frag<DT> f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if constexpr (DT == PSWIN_BF16) {
            f.v[i] = (__bf16)lo[i];
            f.v[4 + i] = (__bf16)hi[i];
        } else {
            f.v[i] = lo[i];
            f.v[4 + i] = hi[i];
        }
    }
    return f;
}

// acc[16x16] += A[16 x 32] . B[32 x 16]; lane (c = lane & 15, g = lane >> 4) supplies row/column c of A/B and the
// 8 contraction elements of its group g.  bf16: one v_mfma_f32_16x16x32_bf16.  f32: eight v_mfma_f32_16x16x4_f32,
// step s contracting element s of every group (same pairing on both operands, so any k order is valid).
template <int DT>
__device__ inline f32x4 mma32(const Frag<DT>& a, const Frag<DT>& b, f32x4 acc) {
    if constexpr (DT == PSWIN_BF16) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[s], b.v[s], acc, 0, 0, 0);
        return acc;
    }
}

// ---------------------------------------------------------------------------------------------
// LDS images: row-major [64 rows][COLS] tiles written row-wise (8 or 4 elements per lane) and read
// TRANSPOSED: lane (c, g) receives, for column col0 + c, the 8 rows {R0 + 0..3, R0 + 16 + 0..3}.
// ---------------------------------------------------------------------------------------------
template <int DT, int COLS>
struct LdsImg {
    // row stride in elements: bf16 rows stay dense (tr reads need 8-byte aligned, rows multiple of 8 B);
    // f32 rows get 4 floats of padding so that the 4 lane groups of a column read hit different banks.
    static constexpr int LD = (DT == PSWIN_BF16) ? COLS : COLS + 4;
    static constexpr int BYTES = PADT * LD * (DT == PSWIN_BF16 ? 2 : 4);
};

template <int DT, int COLS>
__device__ inline void lds_write_frag(char* img, int row, int col, const Frag<DT>& f) {
    constexpr int LD = LdsImg<DT, COLS>::LD;
    if constexpr (DT == PSWIN_BF16) {
        *reinterpret_cast<bf16x8*>(img + ((size_t)row * LD + col) * 2) = f.v;
    } else {
        float* p = reinterpret_cast<float*>(img) + (size_t)row * LD + col;
        f32x4 a = {f.v[0], f.v[1], f.v[2], f.v[3]}, b = {f.v[4], f.v[5], f.v[6], f.v[7]};
        *reinterpret_cast<f32x4*>(p) = a;
        *reinterpret_cast<f32x4*>(p + 4) = b;
    }
}

// 4 consecutive elements of one row (an accumulator quad) -> LDS
template <int DT, int COLS>
__device__ inline void lds_write_quad(char* img, int row, int col, f32x4 q) {
    constexpr int LD = LdsImg<DT, COLS>::LD;
    if constexpr (DT == PSWIN_BF16) {
        bf16x4 b = {(__bf16)q[0], (__bf16)q[1], (__bf16)q[2], (__bf16)q[3]};
        *reinterpret_cast<bf16x4*>(img + ((size_t)row * LD + col) * 2) = b;
    } else {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(img) + (size_t)row * LD + col) = q;
    }
}

// transposed operand read.  EXEC must be all ones (ds_read_b64_tr_b16 gathers across the 16-lane group).
template <int DT, int COLS>
__device__ inline Frag<DT> lds_read_tr(const char* img, int R0, int col0, int c, int g) {
    constexpr int LD = LdsImg<DT, COLS>::LD;
    Frag<DT> f;
    if constexpr (DT == PSWIN_BF16) {
        // lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of the 4 x 16 block; it receives
        // column (4q+p) of the 4 rows.
        const int q = c >> 2, p = c & 3;
        typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
        const char* a0 = img + ((size_t)(R0 + q) * LD + col0 + 4 * p) * 2;
        const char* a1 = a0 + (size_t)16 * LD * 2;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a1));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        f.v = __builtin_bit_cast(bf16x8, both);
    } else {
        const float* base = reinterpret_cast<const float*>(img) + col0 + c;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) f.v[jj] = base[(size_t)(R0 + 16 * (jj >> 2) + (jj & 3)) * LD];
    }
    (void)g;
    return f;
}

// 4 consecutive output elements of one row
template <int DT>
__device__ inline void store_quad(void* base, size_t off, f32x4 q) {
    store4<DT>(base, off, q);
}

__device__ inline float group_max(float v) {   // across the 4 lane groups (lanes c, c+16, c+32, c+48)
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
}
__device__ inline float group_sum(float v) {
    v += __shfl_xor(v, 16);
    return v + __shfl_xor(v, 32);
}
__device__ inline float row16_sum(float v) {   // across the 16 lanes of a group
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    return v + __shfl_xor(v, 8);
}

struct AttnArgs {
    const void *q, *k, *v;
    const float* bias;      // fwd: bias_ij, bwd: bias_ji
    void* out;              // fwd: out ; bwd: unused
    const void* dout;       // bwd
    float* lse;             // fwd: written ; bwd: read
    void *dq, *dk, *dv;     // bwd
    float* dbias;           // bwd (may be null)
    int ld_qkv, ld_out, ld_dqkv;
    int nb, heads, reps_per_chunk, n_items;
    float scale;
};

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int DT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void attn_fwd_kernel(AttnArgs a) {
    using VImg = LdsImg<DT, HD>;
    __shared__ __attribute__((aligned(16))) char smem[WAVES * VImg::BYTES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
    const int item = blockIdx.x * WAVES + wave;
    if (item >= a.n_items) return;   // wave-uniform
    const int h = item % a.heads;
    const int wb = (item / a.heads) % a.nb;
    const int chunk = item / (a.heads * a.nb);
    char* vimg = smem + wave * VImg::BYTES;

    // bias tile of (wb, h): query on the lane, 4 consecutive keys per quad
    f32x4 bias[4][4];
    {
        const float* bt = a.bias + ((size_t)wb * a.heads + h) * (PADT * PADT);
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
                bias[ti][tj] = *reinterpret_cast<const f32x4*>(bt + (16 * ti + c) * PADT + 16 * tj + 4 * g);
    }

    for (int r = 0; r < a.reps_per_chunk; ++r) {
        const size_t win = (size_t)(chunk * a.reps_per_chunk + r) * a.nb + wb;
        const size_t row0 = win * TOK;
        Frag<DT> qf[4], kf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = 16 * t + c;
            const bool ok = row < TOK;
            const size_t off = (row0 + row) * (size_t)a.ld_qkv + h * HD + 8 * g;
            qf[t] = scale_frag<DT>(load_frag<DT>(a.q, off, ok), a.scale);
            kf[t] = load_frag<DT>(a.k, off, ok);
            Frag<DT> vf = load_frag<DT>(a.v, off, ok);
            lds_write_frag<DT, HD>(vimg, row, 8 * g, vf);    // rows >= 49 are written as zeros
        }
        // V^T operand fragments: [k-step s][d tile dt]
        Frag<DT> vt[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) vt[s][dt] = lds_read_tr<DT, HD>(vimg, 32 * s + 4 * g, 16 * dt, c, g);

#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            // S^T tile row: keys 16 tj + 4 g + r on the accumulator, query 16 ti + c on the lane
            f32x4 s4[4];
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) s4[tj] = mma32<DT>(kf[tj], qf[ti], bias[ti][tj]);
            float m = -INFINITY;
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int e = 0; e < 4; ++e) m = fmaxf(m, s4[tj][e]);
            m = group_max(m);
            const float mb = m * LOG2E;
            float l = 0.f;
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float p = __builtin_amdgcn_exp2f(s4[tj][e] * LOG2E - mb);
                    s4[tj][e] = p;
                    l += p;
                }
            l = group_sum(l);
            // O^T[d][i] = sum_j V^T[d][j] P^T[j][i]
            f32x4 o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                Frag<DT> pf = pack_frag<DT>(s4[2 * s], s4[2 * s + 1]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) o[dt] = mma32<DT>(vt[s][dt], pf, o[dt]);
            }
            const int i = 16 * ti + c;
            const float inv_l = 1.0f / l;
            if (i < TOK) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    store_quad<DT>(a.out, (row0 + i) * (size_t)a.ld_out + h * HD + 16 * dt + 4 * g, o[dt] * inv_l);
            }
            if (g == 0) a.lse[(win * a.heads + h) * PADT + i] = (i < TOK) ? m + logf(l) : INFINITY;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
template <int DT>
struct BwdLds {
    static constexpr int ROW = LdsImg<DT, HD>::BYTES;      // Q, K, dO images [64][32]
    static constexpr int TT = LdsImg<DT, PADT>::BYTES;     // dS^T image [64 keys][64 queries]
    static constexpr int BYTES = 3 * ROW + TT;
};

template <int DT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void attn_bwd_kernel(AttnArgs a) {
    using L = BwdLds<DT>;
    __shared__ __attribute__((aligned(16))) char smem[WAVES * L::BYTES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
    const int item = blockIdx.x * WAVES + wave;
    if (item >= a.n_items) return;   // wave-uniform
    const int h = item % a.heads;
    const int wb = (item / a.heads) % a.nb;
    const int chunk = item / (a.heads * a.nb);
    char* qimg = smem + wave * L::BYTES;
    char* kimg = qimg + L::ROW;
    char* doimg = kimg + L::ROW;
    char* timg = doimg + L::ROW;
    const float* bt = a.bias + ((size_t)wb * a.heads + h) * (PADT * PADT);   // bias_ji: [key j][query i]

    f32x4 gsum[4][4];   // sum over the batch loop of dS: [ti][tj], rows i = 16 ti + 4 g + e, column j = 16 tj + c
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) gsum[ti][tj] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int r = 0; r < a.reps_per_chunk; ++r) {
        const size_t win = (size_t)(chunk * a.reps_per_chunk + r) * a.nb + wb;
        const size_t row0 = win * TOK;
        Frag<DT> qf[4], kf[4], vf[4], dof[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = 16 * t + c;
            const bool ok = row < TOK;
            const size_t off = (row0 + row) * (size_t)a.ld_qkv + h * HD + 8 * g;
            qf[t] = scale_frag<DT>(load_frag<DT>(a.q, off, ok), a.scale);
            kf[t] = load_frag<DT>(a.k, off, ok);
            vf[t] = load_frag<DT>(a.v, off, ok);
            dof[t] = load_frag<DT>(a.dout, (row0 + row) * (size_t)a.ld_out + h * HD + 8 * g, ok);
            lds_write_frag<DT, HD>(qimg, row, 8 * g, qf[t]);
            lds_write_frag<DT, HD>(kimg, row, 8 * g, kf[t]);
            lds_write_frag<DT, HD>(doimg, row, 8 * g, dof[t]);
        }
        f32x4 dv[2][4], dk[2][4];   // [dt][tj]: rows d = 16 dt + 4 g + e, column key j = 16 tj + c
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) {
                dv[dt][tj] = f32x4{0.f, 0.f, 0.f, 0.f};
                dk[dt][tj] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        const float* lse_row = a.lse + (win * a.heads + h) * PADT;

#pragma unroll
        for (int s = 0; s < 2; ++s) {          // a pair of query tiles = one 32-deep contraction step for dV, dK
            f32x4 p4[2][4], ds4[2][4];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int ti = 2 * s + tt;
                const f32x4 lse4 = *reinterpret_cast<const f32x4*>(lse_row + 16 * ti + 4 * g);
                f32x4 delta = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bt + (16 * tj + c) * PADT + 16 * ti + 4 * g);
                    f32x4 sc = mma32<DT>(qf[ti], kf[tj], b4);                       // S[i][j] + bias
                    f32x4 dp = mma32<DT>(dof[ti], vf[tj], f32x4{0.f, 0.f, 0.f, 0.f});   // dP[i][j]
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float p = __builtin_amdgcn_exp2f((sc[e] - lse4[e]) * LOG2E);
                        p4[tt][tj][e] = p;
                        delta[e] += p * dp[e];
                    }
                    ds4[tt][tj] = dp;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) delta[e] = row16_sum(delta[e]);
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) ds4[tt][tj][e] = p4[tt][tj][e] * (ds4[tt][tj][e] - delta[e]);
                    gsum[ti][tj] = gsum[ti][tj] + ds4[tt][tj];
                    lds_write_quad<DT, PADT>(timg, 16 * tj + c, 16 * ti + 4 * g, ds4[tt][tj]);   // dS^T[j][i..i+3]
                }
            }
            // dV^T[d][j] += dO^T[d][i] P[i][j] ; dK^T[d][j] += Q^T[d][i] dS[i][j]   (i over this tile pair)
            Frag<DT> dot[2], qt[2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dot[dt] = lds_read_tr<DT, HD>(doimg, 32 * s + 4 * g, 16 * dt, c, g);
                qt[dt] = lds_read_tr<DT, HD>(qimg, 32 * s + 4 * g, 16 * dt, c, g);
            }
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) {
                Frag<DT> pf = pack_frag<DT>(p4[0][tj], p4[1][tj]);
                Frag<DT> dsf = pack_frag<DT>(ds4[0][tj], ds4[1][tj]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt][tj] = mma32<DT>(dot[dt], pf, dv[dt][tj]);
                    dk[dt][tj] = mma32<DT>(qt[dt], dsf, dk[dt][tj]);
                }
            }
        }
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
            const int j = 16 * tj + c;
            if (j < TOK) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const size_t off = (row0 + j) * (size_t)a.ld_dqkv + h * HD + 16 * dt + 4 * g;
                    store_quad<DT>(a.dv, off, dv[dt][tj]);
                    store_quad<DT>(a.dk, off, dk[dt][tj]);
                }
            }
        }
        // dQ^T[d][i] = scale * sum_j K^T[d][j] dS^T[j][i]
        f32x4 dq[2][4];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) dq[dt][ti] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            Frag<DT> kt[2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) kt[dt] = lds_read_tr<DT, HD>(kimg, 32 * s + 4 * g, 16 * dt, c, g);
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
                Frag<DT> tf = lds_read_tr<DT, PADT>(timg, 32 * s + 4 * g, 16 * ti, c, g);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) dq[dt][ti] = mma32<DT>(kt[dt], tf, dq[dt][ti]);
            }
        }
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            const int i = 16 * ti + c;
            if (i < TOK) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    store_quad<DT>(a.dq, (row0 + i) * (size_t)a.ld_dqkv + h * HD + 16 * dt + 4 * g,
                                   dq[dt][ti] * a.scale);
            }
        }
    }
    if (a.dbias) {
        float* gt = a.dbias + (((size_t)chunk * a.nb + wb) * a.heads + h) * (PADT * PADT);
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
                *reinterpret_cast<f32x4*>(gt + (16 * tj + c) * PADT + 16 * ti + 4 * g) = gsum[ti][tj];
    }
}

// ---------------------------------------------------------------------------------------------
// bias tiles and their adjoint
// ---------------------------------------------------------------------------------------------
__device__ inline int rel_index(int i, int j) {
    return (i / PSWIN_WS - j / PSWIN_WS + PSWIN_WS - 1) * (2 * PSWIN_WS - 1) + (i % PSWIN_WS - j % PSWIN_WS + PSWIN_WS - 1);
}

__global__ void bias_build_kernel(const float* __restrict__ dist, int n_dist, const float* __restrict__ alpha,
                                  const float* __restrict__ beta, const float* __restrict__ mask, int n_mask,
                                  int heads, float* __restrict__ bias_ij, float* __restrict__ bias_ji) {
    const int wb = blockIdx.x;
    for (int e = threadIdx.x; e < PADT * PADT; e += blockDim.x) {
        const int i = e / PADT, j = e - i * PADT;
        const bool real = i < TOK && j < TOK;
        float d = 0.f, mk = 0.f;
        int idx = 0;
        if (real) {
            idx = rel_index(i, j);
            if (dist) d = dist[((size_t)(wb % n_dist) * TOK + i) * TOK + j];
            if (mask) mk = mask[((size_t)(wb % n_mask) * TOK + i) * TOK + j];
        }
        for (int h = 0; h < heads; ++h) {
            float val;
            if (real) {
                // same rounding sequence as the reference: (d * alpha + beta) [+ mask]   (HOT:255-256, 294, 301)
                val = beta[idx * heads + h];
                if (dist) val = __fadd_rn(__fmul_rn(d, alpha[idx * heads + h]), val);
                if (mask) val = __fadd_rn(val, mk);
            } else {
                val = (j >= TOK) ? -INFINITY : 0.f;
            }
            const size_t tile = ((size_t)wb * heads + h) * (PADT * PADT);
            bias_ij[tile + i * PADT + j] = val;
            if (bias_ji) bias_ji[tile + j * PADT + i] = val;
        }
    }
}

constexpr int NBINS = (2 * PSWIN_WS - 1) * (2 * PSWIN_WS - 1);   // 169
constexpr int BIAS_BWD_BLOCKS = 128;
constexpr int BIAS_BWD_THREADS = 256;
constexpr int BIAS_BWD_EPT = (TOK * TOK + BIAS_BWD_THREADS - 1) / BIAS_BWD_THREADS;   // 10 (i, j) pairs per thread

// Stage 1: per block and head, sum over a strided subset of the tiles of g and g * dist for every (i, j) pair.
// No atomics: a thread owns the same 10 pairs for every tile, so the sums stay in registers.
// partial[block][h][2][49*49]
__global__ __launch_bounds__(BIAS_BWD_THREADS) void bias_bwd_partial_kernel(
    const float* __restrict__ dbias_ji, int n_tiles, int nb, const float* __restrict__ dist, int n_dist, int heads,
    float* __restrict__ partial) {
    const int h = blockIdx.y;
    float sb[BIAS_BWD_EPT], sa[BIAS_BWD_EPT];
#pragma unroll
    for (int k = 0; k < BIAS_BWD_EPT; ++k) sb[k] = sa[k] = 0.f;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const float* gt = dbias_ji + ((size_t)tile * heads + h) * (PADT * PADT);
        const float* dt = dist ? dist + (size_t)((tile % nb) % n_dist) * TOK * TOK : nullptr;
#pragma unroll
        for (int k = 0; k < BIAS_BWD_EPT; ++k) {
            const int e = threadIdx.x + k * BIAS_BWD_THREADS;      // e = j * 49 + i: i contiguous in the ji tile
            if (e < TOK * TOK) {
                const int j = e / TOK, i = e - j * TOK;
                const float gval = gt[j * PADT + i];
                sb[k] += gval;
                if (dt) sa[k] += gval * dt[i * TOK + j];
            }
        }
    }
    float* out = partial + ((size_t)blockIdx.x * heads + h) * 2 * TOK * TOK;
#pragma unroll
    for (int k = 0; k < BIAS_BWD_EPT; ++k) {
        const int e = threadIdx.x + k * BIAS_BWD_THREADS;
        if (e < TOK * TOK) {
            out[e] = sb[k];
            out[TOK * TOK + e] = sa[k];
        }
    }
}

// Stage 2 (after the column sum over blocks): one thread per (table entry, head) adds the <= 49 pairs of its entry.
// summed: [heads][2][49*49]
__global__ void bias_bwd_final_kernel(const float* __restrict__ summed, int heads, bool has_dist,
                                      float* __restrict__ dalpha, float* __restrict__ dbeta) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;   // over 169 * heads
    if (t >= NBINS * heads) return;
    const int idx = t / heads, h = t - idx * heads;
    const int dh = idx / (2 * PSWIN_WS - 1) - (PSWIN_WS - 1), dw = idx % (2 * PSWIN_WS - 1) - (PSWIN_WS - 1);
    const float* p = summed + (size_t)h * 2 * TOK * TOK;
    float sb = 0.f, sa = 0.f;
    for (int hj = 0; hj < PSWIN_WS; ++hj) {        // pairs with hi - hj = dh and wi - wj = dw
        const int hi = hj + dh;
        if (hi < 0 || hi >= PSWIN_WS) continue;
        for (int wj = 0; wj < PSWIN_WS; ++wj) {
            const int wi = wj + dw;
            if (wi < 0 || wi >= PSWIN_WS) continue;
            const int e = (hj * PSWIN_WS + wj) * TOK + hi * PSWIN_WS + wi;   // j * 49 + i
            sb += p[e];
            if (has_dist) sa += p[TOK * TOK + e];
        }
    }
    dbeta[t] = sb;
    if (dalpha) dalpha[t] = sa;
}

inline int pick_chunks(int reps, int nb, int heads) {
    // enough independent waves to fill 256 CUs x 8 waves, while keeping the batch loop (bias reuse) long
    const long long target = 4096;
    int best = reps;
    for (int ch = 1; ch <= reps; ++ch) {
        if (reps % ch) continue;
        if ((long long)ch * nb * heads >= target) {
            best = ch;
            break;
        }
    }
    return best;
}

inline int check_attn_common(const void* q, const void* k, const void* v, int ld_qkv, int n_windows, int nb,
                             int heads, int dtype) {
    PSWIN_CHECK_ARG(q && k && v);
    PSWIN_CHECK_ARG(valid_dtype(dtype));
    PSWIN_CHECK_ARG(n_windows > 0 && nb > 0 && heads > 0 && n_windows % nb == 0);
    PSWIN_CHECK_ARG(ld_qkv >= heads * HD && ld_qkv % 8 == 0);
    PSWIN_CHECK_ARG(aligned16(q) && aligned16(k) && aligned16(v));
    PSWIN_CHECK_ARG((long long)n_windows * TOK * ld_qkv < (1ll << 40));
    return PSWIN_OK;
}

}  // namespace

extern "C" int pswin_attn_bias_build(const float* dist, int n_dist, const float* alpha, const float* beta,
                                     const float* mask, int n_mask, int n_bias_windows, int heads, float* bias_ij,
                                     float* bias_ji, void* stream) {
    PSWIN_CHECK_ARG(beta && bias_ij && n_bias_windows > 0 && heads > 0);
    PSWIN_CHECK_ARG(!dist || (alpha && n_dist > 0 && n_bias_windows % n_dist == 0));
    PSWIN_CHECK_ARG(!mask || (n_mask > 0 && n_bias_windows % n_mask == 0));
    hipLaunchKernelGGL(bias_build_kernel, dim3(n_bias_windows), dim3(256), 0, (hipStream_t)stream, dist,
                       dist ? n_dist : 1, alpha, beta, mask, mask ? n_mask : 1, heads, bias_ij, bias_ji);
    PSWIN_LAUNCH_RET();
}

extern "C" int pswin_attn_fwd(const void* q, const void* k, const void* v, int ld_qkv, const float* bias_ij,
                              void* out, int ld_out, float* lse, int n_windows, int n_bias_windows, int heads,
                              float scale, int dtype, void* stream) {
    int rc = check_attn_common(q, k, v, ld_qkv, n_windows, n_bias_windows, heads, dtype);
    if (rc) return rc;
    PSWIN_CHECK_ARG(bias_ij && out && lse && aligned16(out) && aligned16(bias_ij));
    PSWIN_CHECK_ARG(ld_out >= heads * HD && ld_out % 8 == 0);
    const int reps = n_windows / n_bias_windows;
    const int chunks = pick_chunks(reps, n_bias_windows, heads);
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v; a.bias = bias_ij; a.out = out; a.lse = lse;
    a.ld_qkv = ld_qkv; a.ld_out = ld_out;
    a.nb = n_bias_windows; a.heads = heads; a.reps_per_chunk = reps / chunks;
    a.n_items = chunks * n_bias_windows * heads;
    a.scale = scale;
    if (dtype == PSWIN_BF16) {
        constexpr int W = 4;
        hipLaunchKernelGGL((attn_fwd_kernel<PSWIN_BF16, W>), dim3((a.n_items + W - 1) / W), dim3(64 * W), 0,
                           (hipStream_t)stream, a);
    } else {
        constexpr int W = 4;
        hipLaunchKernelGGL((attn_fwd_kernel<PSWIN_F32, W>), dim3((a.n_items + W - 1) / W), dim3(64 * W), 0,
                           (hipStream_t)stream, a);
    }
    PSWIN_LAUNCH_RET();
}

extern "C" int pswin_attn_bwd(const void* q, const void* k, const void* v, int ld_qkv, const float* bias_ji,
                              const void* dout, int ld_out, const float* lse, void* dq, void* dk, void* dv,
                              int ld_dqkv, float* dbias_ji, int n_chunks, int n_windows, int n_bias_windows,
                              int heads, float scale, int dtype, void* stream) {
    int rc = check_attn_common(q, k, v, ld_qkv, n_windows, n_bias_windows, heads, dtype);
    if (rc) return rc;
    PSWIN_CHECK_ARG(bias_ji && dout && lse && dq && dk && dv);
    PSWIN_CHECK_ARG(aligned16(dout) && aligned16(dq) && aligned16(dk) && aligned16(dv) && aligned16(bias_ji) &&
                    aligned16(lse) && aligned16(dbias_ji));
    PSWIN_CHECK_ARG(ld_out >= heads * HD && ld_out % 8 == 0 && ld_dqkv >= heads * HD && ld_dqkv % 8 == 0);
    const int reps = n_windows / n_bias_windows;
    PSWIN_CHECK_ARG(n_chunks >= 1 && n_chunks <= reps && reps % n_chunks == 0);
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v; a.bias = bias_ji; a.dout = dout; a.lse = const_cast<float*>(lse);
    a.dq = dq; a.dk = dk; a.dv = dv; a.dbias = dbias_ji;
    a.ld_qkv = ld_qkv; a.ld_out = ld_out; a.ld_dqkv = ld_dqkv;
    a.nb = n_bias_windows; a.heads = heads; a.reps_per_chunk = reps / n_chunks;
    a.n_items = n_chunks * n_bias_windows * heads;
    a.scale = scale;
    if (dtype == PSWIN_BF16) {
        constexpr int W = 4;
        hipLaunchKernelGGL((attn_bwd_kernel<PSWIN_BF16, W>), dim3((a.n_items + W - 1) / W), dim3(64 * W), 0,
                           (hipStream_t)stream, a);
    } else {
        constexpr int W = 3;
        hipLaunchKernelGGL((attn_bwd_kernel<PSWIN_F32, W>), dim3((a.n_items + W - 1) / W), dim3(64 * W), 0,
                           (hipStream_t)stream, a);
    }
    PSWIN_LAUNCH_RET();
}

extern "C" int pswin_attn_suggest_chunks(int n_windows, int n_bias_windows, int heads) {
    if (n_windows <= 0 || n_bias_windows <= 0 || heads <= 0 || n_windows % n_bias_windows) return PSWIN_ERR_ARG;
    return pick_chunks(n_windows / n_bias_windows, n_bias_windows, heads);
}

extern "C" int pswin_attn_bias_bwd_workspace(int heads) {
    return heads > 0 ? (BIAS_BWD_BLOCKS + 1) * heads * 2 * TOK * TOK : PSWIN_ERR_ARG;
}

extern "C" int pswin_attn_bias_bwd(const float* dbias_ji, int n_tiles, int n_bias_windows, const float* dist,
                                   int n_dist, int heads, float* dalpha, float* dbeta, float* workspace,
                                   void* stream) {
    PSWIN_CHECK_ARG(dbias_ji && dbeta && workspace && n_tiles > 0 && n_bias_windows > 0 && heads > 0);
    PSWIN_CHECK_ARG(n_tiles % n_bias_windows == 0);
    PSWIN_CHECK_ARG(!dist || (n_dist > 0 && n_bias_windows % n_dist == 0));
    const int blocks = n_tiles < BIAS_BWD_BLOCKS ? n_tiles : BIAS_BWD_BLOCKS;
    hipLaunchKernelGGL(bias_bwd_partial_kernel, dim3(blocks, heads), dim3(BIAS_BWD_THREADS), 0, (hipStream_t)stream,
                       dbias_ji, n_tiles, n_bias_windows, dist, dist ? n_dist : 1, heads, workspace);
    const int ncol = heads * 2 * TOK * TOK;
    float* summed = workspace + (size_t)BIAS_BWD_BLOCKS * ncol;
    launch_colsum(workspace, blocks, ncol, summed, (hipStream_t)stream);
    hipLaunchKernelGGL(bias_bwd_final_kernel, dim3((NBINS * heads + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       summed, heads, dist != nullptr, dist ? dalpha : nullptr, dbeta);
    PSWIN_LAUNCH_RET();
}
