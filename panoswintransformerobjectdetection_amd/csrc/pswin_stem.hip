// PatchEmbed stem of PanoSwin (HOT:742-750), fused for gfx950, bf16 operands / f32 accumulation.
//
//   x [B,3,H,W] -> Conv3x3(3->32) -> BN -> ReLU -> Conv3x3(32->64) -> BN -> ReLU -> Conv4x4/s4(64->96) -> tokens
//
// The reference runs the two 3x3 convolutions at FULL input resolution, so their activations are by far the largest
// tensors of the network (268 MB and 537 MB in bf16 at B = 8, 512x1024); with library convolutions + separate
// BatchNorm passes the stem moves ~10 GB per training step (4.7 ms of an 18 ms step, measured).  Here only ONE
// full-resolution tensor exists in each direction: y2 = conv2 output (forward), dy2 = its gradient (backward).
//   * conv1 (27 MACs per output) is never stored: every kernel that needs a1 = relu(bn1(conv1 x)) recomputes it
//     from the 8-byte-per-pixel input tile with three K=16 MFMAs per 16 pixels;
//   * BatchNorm statistics are accumulated inside the producing kernel (conv1: a statistics-only pass; conv2: in
//     the epilogue), BN + ReLU are applied when the consumer loads its operand (conv3 reads y2);
//   * backward: d(tokens) -> [conv3 data gradient + BN2 backward] -> dy2 in two passes over y2 (sums, then apply);
//     conv2 weight gradient and conv2 data gradient read dy2; the conv2 data gradient kernel also finishes BN1's
//     backward and conv1's weight gradient analytically: dW1 = rstd g (G - m1 X1 - m2 Y) with G = sum g1 (x) xp
//     accumulated in-kernel, and X1, Y = rstd (W1 XX - mean X1) from the input autocorrelation XX = sum xp (x) xp
//     that the forward statistics pass produced -- g1 (268 MB) is never written.
// Convolution biases in front of a BatchNorm cancel and are not applied (they only shift the tracked mean: host side).
//
// Layouts: x4 [B][H][W][4] bf16 (channel 3 = 1.0: the "ones" slot that makes XX carry the plain sums X1 and the
// pixel count); y2, dy2 [B][H][W][64] bf16; tokens [B*H/4*W/4][96] bf16.
// Weights are repacked by the caller (tiny): w1p [32][12 taps][4] (taps 9..11 and channel 3 zero), w2p [9][64 out][32 in],
// w2t [9][32 in][64 out], w3p [16][96 out][64 in], w3t [16][64 in][96 out], all bf16.
#include "pswin_common.hpp"

using namespace pswin;

namespace {

constexpr int C1 = 32, C2 = 64, C3 = 96;
constexpr int TW = 32;                       // tile width of the 3x3 kernels (two 16-pixel MFMA column groups)
constexpr int TWG = 512, TNW = 8;            // the token kernels (conv3): 8 waves x 16 tokens
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef unsigned long long u64;
using rsrc_t = __amdgpu_buffer_rsrc_t;

__device__ inline f32x4 mfma16(s16x4 a, s16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
__device__ inline f32x4 mfma32(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// LDS images with power-of-two rows, 16-byte chunks XOR-swizzled by the row so that both the row-wise 16-byte accesses
// of 8 consecutive rows and the transposed 8-byte reads (8 rows x 32 bytes per half wave) are bank-conflict free
__device__ inline int off64(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 3)) << 4); }    // 32 bf16 / row
__device__ inline int off128(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }         // 64 bf16 / row

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
// transposed operand fragment: lane (q = c >> 2, p = c & 3) of each 16-lane group passes the address of 8 bytes
// (4 columns 4p..4p+3) of k-row q (lo) / q + 4 (hi); lane c receives column c of the 8 rows
__device__ inline bf16x8 tr_pair(const char* lo, const char* hi) {
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lo));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(hi));
    s16x8 both = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, both);
}

template <int CTRL>
__device__ inline float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ inline float row16_sum(float v) {   // over the 16 lanes of a group
    v = dpp_add<0xB1>(v);
    v = dpp_add<0x4E>(v);
    v = dpp_add<0x141>(v);
    return dpp_add<0x140>(v);
}
// v_permlane16_swap: exchanges the odd 16-lane rows of a with the even rows of b.  The builtin (not inline asm) so that
// the compiler's hazard recognizer sees the instruction: its operands often come straight from MFMA accumulators, and
// the MFMA-write -> VALU-read wait states are software managed (an asm version with a fixed s_nop read stale values).
__device__ inline void swap16_u32(unsigned& a, unsigned& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}

// One v_cvt_pk_bf16_f32 per pair (pack2_bf16).  Round 2 kept the two-conversion form here because the short form produced non-finite
// weight gradients; round 3 found why (it was never the conversion): the store of stem_conv3_bwd_kernel<true> below -- see the comment
// there.  PSWIN_STEM_PACK2=0 builds the old form (A/B).
#ifndef PSWIN_STEM_PACK2
#define PSWIN_STEM_PACK2 1
#endif
__device__ inline unsigned pack_bf16(float lo, float hi) {
#if PSWIN_STEM_PACK2
    return pack2_bf16(lo, hi);
#else
    return (unsigned)f32_to_bf16_bits(lo) | ((unsigned)f32_to_bf16_bits(hi) << 16);
#endif
}
__device__ inline float bf_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ inline float bf_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// A lane (c, g) holds for one pixel / token the accumulator quads q0 = ch[4g..4g+3] and q1 = ch[16+4g..16+4g+3] of a
// 32-channel group; after exchanging q1 of the even groups with q0 of the odd ones every lane owns 8 consecutive
// channels starting at d0 = 8 (g >> 1) + 16 (g & 1): returned packed as 4 dwords of bf16 pairs.
__device__ inline u32x4 pack_row8(int g, f32x4 q0, f32x4 q1) {
    unsigned a0 = pack_bf16(q0[0], q0[1]), a1 = pack_bf16(q0[2], q0[3]);
    unsigned b0 = pack_bf16(q1[0], q1[1]), b1 = pack_bf16(q1[2], q1[3]);
    swap16_u32(a0, b0);
    swap16_u32(a1, b1);
    (void)g;
    return u32x4{a0, a1, b0, b1};
}
// the same exchange on f32 quads (8 consecutive channels as two quads).  Whole-vector bit casts only: hipcc (ROCm 7.2)
// folds __builtin_bit_cast(T, vec[e]) inside an unrolled loop to element 0.
__device__ inline void exchange_row8(f32x4& q0, f32x4& q1) {
    const u32x4 a = __builtin_bit_cast(u32x4, q0), b = __builtin_bit_cast(u32x4, q1);
    unsigned a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
    swap16_u32(a0, b0);
    swap16_u32(a1, b1);
    swap16_u32(a2, b2);
    swap16_u32(a3, b3);
    q0 = __builtin_bit_cast(f32x4, u32x4{a0, a1, a2, a3});
    q1 = __builtin_bit_cast(f32x4, u32x4{b0, b1, b2, b3});
}
__device__ inline int row8_d0(int g) { return 8 * (g >> 1) + 16 * (g & 1); }

// ---------------------------------------------------------------------------------------------
// input repack: [B,3,H,W] f32 -> [B,H,W,4] bf16, channel 3 = 1
// ---------------------------------------------------------------------------------------------
__global__ void stem_pack_kernel(const float* __restrict__ x, long long npix_per_img, long long total, u64* __restrict__ x4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long b = i / npix_per_img, p = i - b * npix_per_img;
    const float* src = x + b * 3 * npix_per_img + p;
    const unsigned lo = pack_bf16(src[0], src[npix_per_img]);
    const unsigned hi = pack_bf16(src[2 * npix_per_img], 1.0f);
    x4[i] = (u64)lo | ((u64)hi << 32);
}

// ---------------------------------------------------------------------------------------------
// F3: tokens = conv3(relu(bn2(y2))) + bias     (4x4 stride 4: one GEMM row per token, K = 16 taps x 64 channels)
// ---------------------------------------------------------------------------------------------
constexpr int W3S_BYTES = C3 * 128;          // one tap slice [96 out][64 in]
constexpr int TOK_WG = TNW * 16;              // 128 tokens per workgroup and token tile
#ifndef PSWIN_STEM_CONV3_TT
#define PSWIN_STEM_CONV3_TT 4                 // token tiles per wave in the conv3 forward kernel (A/B builds: 1, 2)
#endif

struct TokGeo {
    long long pix;       // pixel index of the token's top-left pixel in [B][H][W]
};
__device__ inline long long token_pix(long long tok, int Hh, int Wh, int H, int W) {
    const long long b = tok / ((long long)Hh * Wh);
    const int rem = (int)(tok - b * Hh * Wh);
    const int ty = rem / Wh, tx = rem - ty * Wh;
    return (b * H + 4 * ty) * (long long)W + 4 * tx;
}

// y -> relu(y * sc + sh) on 8 packed bf16 channels
__device__ inline bf16x8 bn_relu8(u32x4 raw, const float (&sc)[8], const float (&sh)[8]) {
    u32x4 o;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const float lo = fmaxf(__builtin_fmaf(bf_lo(raw[d]), sc[2 * d], sh[2 * d]), 0.f);
        const float hi = fmaxf(__builtin_fmaf(bf_hi(raw[d]), sc[2 * d + 1], sh[2 * d + 1]), 0.f);
        o[d] = pack_bf16(lo, hi);
    }
    return __builtin_bit_cast(bf16x8, o);
}

// TT token tiles (16 tokens each) per wave and weight stage: a tap's 12 KB weight slice is staged once per TT * 128 tokens of the
// workgroup and every weight fragment read from LDS feeds TT MFMAs.  With TT = 1 (the first version) a wave had ONE tile: 12 LDS
// fragment reads for 12 MFMAs (twice what the LDS pipe delivers under a busy matrix pipe), a barrier and an exposed memory round trip
// per tap and 128 tokens -- 175 us for a pass whose traffic takes 93.
template <int TT>
__global__ __launch_bounds__(TWG) void stem_conv3_fwd_kernel(const void* __restrict__ y2, const float* __restrict__ scale2,
                                                            const float* __restrict__ shift2, const void* __restrict__ w3p,
                                                            const float* __restrict__ bias3, int H, int W, long long M,
                                                            long long y2_bytes, void* __restrict__ tok_out) {
    __shared__ __attribute__((aligned(16))) char w3s[2][W3S_BYTES];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
    const int Hh = H / 4, Wh = W / 4;
    const long long tok_wg = (long long)blockIdx.x * (TOK_WG * TT);
    // tile t of this wave: tokens tok_wg + (TT * wave + t) * 16 + c
    long long tok[TT];
    bool valid[TT];
    const long long base_pix = token_pix(tok_wg, Hh, Wh, H, W);                 // uniform
    const long long rem_bytes = y2_bytes - base_pix * 128;
    const rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(y2)) + base_pix * 128, 0,
                                                        (int)(rem_bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : rem_bytes), 0x00020000);
    unsigned voff[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        tok[t] = tok_wg + (TT * wave + t) * 16 + c;
        valid[t] = tok[t] < M;
        if (!valid[t]) tok[t] = M - 1;
        voff[t] = (unsigned)((token_pix(tok[t], Hh, Wh, H, W) - base_pix) * 128) + 16u * g;
    }
    float sc[2][8], sh[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sc[s][j] = scale2[32 * s + 8 * g + j];
            sh[s][j] = shift2[32 * s + 8 * g + j];
        }
    // weight slices: 768 16-byte chunks per tap, staged through registers into the other LDS buffer
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(w3p);
    const int ch0 = threadIdx.x, ch1 = threadIdx.x + TWG;
    auto stage_load = [&](int tap, u32x4& r0, u32x4& r1) {
        r0 = wsrc[tap * 768 + ch0];
        if (ch1 < 768) r1 = wsrc[tap * 768 + ch1];
    };
    auto stage_store = [&](int buf, const u32x4& r0, const u32x4& r1) {
        *reinterpret_cast<u32x4*>(w3s[buf] + off128(ch0 >> 3, ch0 & 7)) = r0;
        if (ch1 < 768) *reinterpret_cast<u32x4*>(w3s[buf] + off128(ch1 >> 3, ch1 & 7)) = r1;
    };
    u32x4 r0, r1 = {0u, 0u, 0u, 0u};
    stage_load(0, r0, r1);
    stage_store(0, r0, r1);
    u32x4 yb[TT][2], yn[TT][2];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        yb[t][0] = __builtin_amdgcn_raw_buffer_load_b128(ys, voff[t], 0, 0);
        yb[t][1] = __builtin_amdgcn_raw_buffer_load_b128(ys, voff[t] + 64u, 0, 0);
    }
    f32x4 acc[TT][6];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int tap = 0; tap < 16; ++tap) {
        const int buf = tap & 1;
        if (tap + 1 < 16) {
            stage_load(tap + 1, r0, r1);
            const int nt = tap + 1;
            const int soff = ((nt >> 2) * W + (nt & 3)) * 128;
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                yn[t][0] = __builtin_amdgcn_raw_buffer_load_b128(ys, voff[t], soff, 0);
                yn[t][1] = __builtin_amdgcn_raw_buffer_load_b128(ys, voff[t] + 64u, soff, 0);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 b[TT];
#pragma unroll
            for (int t = 0; t < TT; ++t) b[t] = bn_relu8(yb[t][s], sc[s], sh[s]);
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(w3s[buf] + off128(16 * mt + c, 4 * s + g));
#pragma unroll
                for (int t = 0; t < TT; ++t) acc[t][mt] = mfma32(a, b[t], acc[t][mt]);
            }
        }
        if (tap + 1 < 16) {
            stage_store(buf ^ 1, r0, r1);
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                yb[t][0] = yn[t][0];
                yb[t][1] = yn[t][1];
            }
        }
        __syncthreads();
    }
    // + bias, bf16, 8 consecutive channels per lane
    float bv[6][4];
#pragma unroll
    for (int mt = 0; mt < 6; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[mt][e] = bias3[16 * mt + 4 * g + e];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
#pragma unroll
        for (int mt = 0; mt < 6; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][mt][e] += bv[mt][e];
        char* dst = reinterpret_cast<char*>(tok_out) + tok[t] * (C3 * 2) + row8_d0(g) * 2;
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
            const u32x4 v = pack_row8(g, acc[t][2 * pr], acc[t][2 * pr + 1]);
            if (valid[t]) *reinterpret_cast<u32x4*>(dst + pr * 64) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// B1 / B2: g2 = (d tokens . W3) [relu(bn2 y2) > 0] per pixel (conv3 data gradient: every pixel belongs to exactly one
// token and tap), then BN2's backward.  APPLY = false: per-channel sum g2, sum g2 * yhat2 (also dbeta2, dgamma2).
// APPLY = true: dy2 = k1 g2 - P y2 - Q  (= gamma rstd (g2 - mean g2 - yhat2 mean(g2 yhat2)) with the means folded).
// prm: f32 [5][64] = scale2, shift2, then (a = rstd, b = -mean rstd, unused) or (k1, P, Q).
// ---------------------------------------------------------------------------------------------
constexpr int W3T_LD = 208;                    // 96 bf16 + 16 bytes: 8 rows x 16 bytes tile the 32 banks
constexpr int W3T_BYTES = C2 * W3T_LD;
constexpr int PART3 = 2 * C2;

// Round 4: persistent, one workgroup per CU, no barrier inside the tap loop.  The round-1 form gave every 128-token group a workgroup
// of its own that walked the 16 taps with a weight slice staged per tap (a barrier and a one-step-ahead prefetch per tap): with one
// workgroup resident per CU each tap cost one HBM round trip (1.8 us; 234 us for a pass whose traffic takes 110).  Now HALF of the taps'
// weight slices (8 x 13 KB = 104 KB) are resident in LDS at a time, a workgroup streams ALL its token groups past them, then loads the
// other half and streams again (d tokens re-read: 192 B against 2 KB of y2 per token) -- two barriers per workgroup lifetime; the 8 waves
// run free, each on its own 16 tokens, with the y2 rows of the next four taps in flight in a register ring that runs on into the wave's
// next token group.
constexpr int W3H_TAPS = 8;
constexpr int W3H_BYTES = W3H_TAPS * W3T_BYTES;            // 106,496

__device__ inline rsrc_t uniform_rsrc(const void* base, long long bytes) {      // a buffer resource the compiler can see is wave-uniform
    const unsigned long long v = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    const int n = (int)(bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : (bytes < 0 ? 0 : bytes));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(n), 0x00020000);
}

template <bool APPLY>
__global__ __launch_bounds__(TWG, 2) void stem_conv3_bwd_kernel(const void* __restrict__ dtok, const void* __restrict__ y2,
                                                               const float* __restrict__ prm, const void* __restrict__ w3t,
                                                               int H, int W, long long M, long long y2_bytes,
                                                               void* __restrict__ dy2, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char w3h[];          // [8 taps][64 in][208 B]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
    const int Hh = H / 4, Wh = W / 4;
    const int d0 = row8_d0(g);
    const long long ngroups = (M + TOK_WG - 1) / TOK_WG;
    // per-lane channels: 32 h + d0 + j
    float p0[2][8], p1[2][8], p2[2][8], p3[2][8], p4[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ch = 32 * h + d0 + j;
            p0[h][j] = prm[ch];
            p1[h][j] = prm[C2 + ch];
            // the statistics pass needs (a, b) of yhat = a y + b only once, at the end: it accumulates sum g y and finishes
            // sum g yhat = a sum g y + b sum g there (32 registers less in the tap loop, which spilled with them)
            p2[h][j] = APPLY ? prm[2 * C2 + ch] : 0.f;
            p3[h][j] = APPLY ? prm[3 * C2 + ch] : 0.f;
            p4[h][j] = APPLY ? prm[4 * C2 + ch] : 0.f;
        }
    float sg[2][8], sgy[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 8; ++j) sg[h][j] = sgy[h][j] = 0.f;

    // one token group of this wave: resources based at the group's first pixel, the lane's token offset, its d-token fragments
    struct Grp {
        rsrc_t ys, ds;
        unsigned voff;
        bool valid;
    };
    auto group_of = [&](long long grp) {
        Grp r;
        const long long tok_wg = grp * TOK_WG;
        long long tok = tok_wg + wave * 16 + c;
        r.valid = grp < ngroups && tok < M;
        if (tok >= M) tok = M - 1;
        const long long base_pix = token_pix(tok_wg < M ? tok_wg : M - 1, Hh, Wh, H, W);
        const long long rem = grp < ngroups ? y2_bytes - base_pix * 128 : 0;
        r.ys = uniform_rsrc(reinterpret_cast<const char*>(y2) + base_pix * 128, rem);
        r.ds = uniform_rsrc(reinterpret_cast<char*>(dy2) + base_pix * 128, APPLY ? rem : 0);
        r.voff = (unsigned)((token_pix(tok, Hh, Wh, H, W) - base_pix) * 128) + 2u * d0;
        return r;
    };
    const rsrc_t dts = uniform_rsrc(dtok, M * (long long)(C3 * 2));
    auto load_bt = [&](long long grp, bf16x8 (&bt)[3]) {         // d tokens as B operands: k = output channel 32 s + 8 g ..
        const long long tok = grp * TOK_WG + wave * 16 + c;
        const unsigned off = (grp < ngroups && tok < M) ? (unsigned)(tok * (C3 * 2) + 16 * g) : 0xFFFFFF00u;       // past the end: zeros
#pragma unroll
        for (int s = 0; s < 3; ++s) bt[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(dts, off == 0xFFFFFF00u ? off : off + 64u * s, 0, 0));
    };
    auto tap_off = [&](int tap) { return ((tap >> 2) * W + (tap & 3)) * 128; };

    for (int half = 0; half < 2; ++half) {
        // this half's eight weight slices -> LDS (12 requests per thread, all in flight together)
        __syncthreads();
        {
            const u32x4* wsrc = reinterpret_cast<const u32x4*>(w3t) + (size_t)half * W3H_TAPS * 768;
            constexpr int WIT = W3H_TAPS * 768 / TWG;
            u32x4 wr[WIT];
#pragma unroll
            for (int it = 0; it < WIT; ++it) wr[it] = wsrc[threadIdx.x + it * TWG];
#pragma unroll
            for (int it = 0; it < WIT; ++it) {
                const int i = threadIdx.x + it * TWG;
                const int tap = i / 768, ch = i - tap * 768;
                *reinterpret_cast<u32x4*>(w3h + tap * W3T_BYTES + (ch / 12) * W3T_LD + (ch % 12) * 16) = wr[it];
            }
        }
        __syncthreads();
        long long grp = blockIdx.x;
        Grp cur = group_of(grp);
        bf16x8 bt[3], btn[3];
        load_bt(grp, bt);
        u32x4 yq[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int so = tap_off(W3H_TAPS * half + u);
            yq[u][0] = __builtin_amdgcn_raw_buffer_load_b128(cur.ys, cur.voff, so, 0);
            yq[u][1] = __builtin_amdgcn_raw_buffer_load_b128(cur.ys, cur.voff + 64u, so, 0);
        }
        for (; grp < ngroups; grp += gridDim.x) {
            const Grp nxt = group_of(grp + gridDim.x);
#pragma unroll
            for (int t8 = 0; t8 < W3H_TAPS; ++t8) {
                const int tap = W3H_TAPS * half + t8;
                // (scheduling barriers at the tap boundaries: without them the compiler hoists the weight-fragment reads of all eight
                // unrolled taps -- 384 registers -- to the top of the loop and spills)
                __builtin_amdgcn_sched_barrier(0);
                if (t8 == 4) load_bt(grp + gridDim.x, btn);         // the next group's d tokens travel with its first y2 rows
                f32x4 acc[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        const bf16x8 a = *reinterpret_cast<const bf16x8*>(w3h + t8 * W3T_BYTES + (16 * mt + c) * W3T_LD + (4 * s + g) * 16);
                        acc[mt] = mfma32(a, bt[s], acc[mt]);
                    }
                }
                // acc[mt][e] = d a2[channel 16 mt + 4 g + e][token c] -> 8 consecutive channels 32 h + d0 .. per lane
                __builtin_amdgcn_sched_barrier(0);
                exchange_row8(acc[0], acc[1]);
                exchange_row8(acc[2], acc[3]);
                const int soff = tap_off(tap);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u32x4 yb = yq[t8 & 3][h];
                    float y[8], v[8];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        y[2 * d] = bf_lo(yb[d]);
                        y[2 * d + 1] = bf_hi(yb[d]);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float da = acc[2 * h + (j >> 2)][j & 3];
                        const bool on = __builtin_fmaf(y[j], p0[h][j], p1[h][j]) > 0.f;
                        const float gg = on ? da : 0.f;
                        if constexpr (APPLY) {
                            v[j] = __builtin_fmaf(p2[h][j], gg, -__builtin_fmaf(p3[h][j], y[j], p4[h][j]));
                        } else {
                            sg[h][j] += gg;
                            sgy[h][j] = __builtin_fmaf(gg, y[j], sgy[h][j]);
                            // pin the update to its tap: left alone, the vectorizer pairs the accumulations of the eight unrolled taps and
                            // sinks them behind the loop body, keeping every tap's 16 accumulator registers alive (120 spilled registers)
                            asm volatile("" : "+v"(sg[h][j]), "+v"(sgy[h][j]));
                        }
                    }
                    if constexpr (APPLY) {
                        const u32x4 o = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
                        // The tap offset rides in the VECTOR offset, not in the scalar one.  hipcc's hazard recognizer pads a 128-bit buffer
                        // store against the next VALU write of its data registers only when the store has NO scalar-register offset (the rule
                        // of the older GCN parts); with `soff` in an SGPR it pads nothing, and the next pixel's arithmetic, which reuses the
                        // four data registers one wait state later, overwrote them before the store had read them -- deterministically wrong
                        // dy2 (f32 bit patterns where bf16 pairs belong) once the packing got short enough (one v_cvt_pk_bf16_f32 per pair) to
                        // put the overwrite that close; DESIGN.md section 5, "the stem's non-finite gradients".
                        if (cur.valid) __builtin_amdgcn_raw_buffer_store_b128(o, cur.ds, cur.voff + 64u * h + (unsigned)soff, 0, 0);
                    }
                }
                // ring: the slot just consumed takes the y2 rows four taps on -- of this group, or of the wave's next group
                {
                    const bool same = t8 + 4 < W3H_TAPS;
                    const int so = tap_off(W3H_TAPS * half + ((t8 + 4) & (W3H_TAPS - 1)));
                    const rsrc_t& rs = same ? cur.ys : nxt.ys;
                    const unsigned vo = same ? cur.voff : nxt.voff;
                    yq[t8 & 3][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0);
                    yq[t8 & 3][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 64u, so, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            cur = nxt;
#pragma unroll
            for (int s = 0; s < 3; ++s) bt[s] = btn[s];
        }
    }
    if constexpr (!APPLY) {
        float* out = partial + ((size_t)blockIdx.x * TNW + wave) * PART3;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ch = 32 * h + d0 + j;
                const float a = row16_sum(sg[h][j]), b = row16_sum(sgy[h][j]);
                if (c == 0) {
                    out[ch] = a;
                    out[C2 + ch] = __builtin_fmaf(prm[2 * C2 + ch], b, prm[3 * C2 + ch] * a);      // sum g yhat = a' sum g y + b' sum g
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// B3: conv3 weight gradient  dW3[out][tap][in] = sum_tokens dtok[token][out] * a2[token, tap][in], a2 = relu(bn2 y2).
// Workgroup = one tap row dy and a range of 32-token contraction steps; wave = (tap dx, input-channel half).
// Round 4 (end): a step's operands -- the raw y2 rows of the 4 taps (16 KB) and the d-token tile (32 rows of 192 B in 256-byte LDS rows)
// -- go global -> LDS by LDS-DMA into a THREE-stage ring, two steps ahead of the MFMAs (counted vmcnt, one raw s_barrier per step, as
// pswin_gemm_tn.hip), and BatchNorm + ReLU are applied to the transposed fragment a lane reads (one channel per lane: two scalars
// instead of 16 registers of coefficients) -- the same values, rounded the same way, as normalising before the LDS store.  Before, the
// rows were staged through registers and a workgroup had ONE step in flight (bytes in flight / latency: 2.8, then 3.5 TB/s with two
// workgroups per CU).  No register load is left inside the loop (hipcc puts s_waitcnt vmcnt(0) in front of a loop-carried load), and all
// LDS fragment reads are inline asm (hipcc orders the ds_read_tr16 builtin behind every LDS-DMA in flight with s_waitcnt vmcnt(0)).
// LDS images: y2 rows off128 (chunk ^ (token & 7), applied on the DMA's source side); d rows: 16 chunks of 16 B, chunk ^ 2 (row & 7),
// chunks 12..15 zero -- conflict-free for the transposed reads of both.
// ---------------------------------------------------------------------------------------------
constexpr int WG3_A2 = 4 * 32 * 128;                // y2 rows of the 4 taps: [dx][32 tokens][64 ch]
constexpr int WG3_DT = 32 * 256;                    // d-token tile
constexpr int WG3_STAGE = WG3_A2 + WG3_DT;          // 24,576 B
constexpr int WG3_ST = 3;                           // ring stages
constexpr int WG3_LDS = WG3_ST * WG3_STAGE;         // 73,728 B (dynamic): two workgroups per CU
constexpr int WG3_TILES = 2 * 6;                    // accumulator tiles per wave
constexpr int WG3_OUT = TNW * WG3_TILES * 256;       // floats per workgroup partial: [wave][nt][mt][e][lane]
typedef __attribute__((address_space(3))) void wg3_lds_void;

template <int OFF>
__device__ inline void wg3_tr(u32x2& dst, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF) : "memory");
}

__global__ __launch_bounds__(TWG, 4) void stem_conv3_wgrad_kernel(const void* __restrict__ dtok, const void* __restrict__ y2,
                                                                 const float* __restrict__ scale2, const float* __restrict__ shift2,
                                                                 int H, int W, int M, int steps_per_wg, long long y2_bytes,
                                                                 float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4, q = c >> 2, p = c & 3;
    const int dx = wave & 3, nh = wave >> 2;
    const int dy = blockIdx.y;
    const int Hh = H / 4, Wh = W / 4;
    // BatchNorm coefficients of this lane's channel per input-channel tile nt: channel 16 (2 nh + nt) + c.  They pass through an asm
    // statement once they are there: otherwise hipcc waits for these two loads INSIDE the loop (a vmcnt that drains the ring)
    float scl[2], shl[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        scl[nt] = scale2[16 * (2 * nh + nt) + c];
        shl[nt] = shift2[16 * (2 * nh + nt) + c];
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(scl[0]), "+v"(scl[1]), "+v"(shl[0]), "+v"(shl[1])::"memory");
    f32x4 acc[2][6];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nsteps = (M + 31) / 32;
    const int s_begin = blockIdx.x * steps_per_wg;
    int s_end = s_begin + steps_per_wg;
    if (s_end > nsteps) s_end = nsteps;

    // wave-uniform buffer resources (readfirstlane: no waterfall loops around the loads)
    auto uni = [](const void* base, long long bytes) {
        const unsigned long long v = (unsigned long long)base;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
    };
    const rsrc_t yr = uni(y2, y2_bytes), dr = uni(dtok, (long long)M * (C3 * 2));
    // LDS-DMA roles per step and wave (three wave instructions):
    //   y2: tap (wave & 3), token blocks 2 (wave >> 2) + {0, 1}; an instruction moves 8 tokens x 128 B, lane i -> token 8 tb + (i >> 3),
    //       physical chunk i & 7 = logical chunk (i & 7) ^ (token & 7);
    //   d:  rows 4 wave .. 4 wave + 3; lane i -> row 4 wave + (i >> 4), physical chunk i & 15 = logical chunk (i & 15) ^ 2 (row & 7);
    //       logical chunks 12..15 and rows past M: out-of-range offset, zeros land in LDS.
    const int pl = lane >> 3, lch = (lane & 7) ^ pl;
    const int drow = 4 * wave + (lane >> 4), dlch = (lane & 15) ^ (2 * (drow & 7));
    auto issue = [&](int st, int stage) {
        char* base = smem + stage * WG3_STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int tb = 2 * (wave >> 2) + j;
            int tok = st * 32 + 8 * tb + pl;
            tok = tok < M ? tok : M - 1;               // rows past M: any valid pixel (their d rows are zeros)
            const int b = tok / (Hh * Wh), rem = tok - b * Hh * Wh;
            const int ty = rem / Wh, tx = rem - ty * Wh;
            const unsigned pix = (unsigned)((b * H + 4 * ty + dy) * W + 4 * tx + (wave & 3));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (wg3_lds_void*)(base + (wave & 3) * 4096 + tb * 1024), 16,
                                                     pix * 128u + 16u * (unsigned)lch, 0, 0, 0);
        }
        const int tok = st * 32 + drow;
        const bool in = (tok < M) & (dlch < 12);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(dr, (wg3_lds_void*)(base + WG3_A2 + wave * 1024), 16,
                                                 in ? (unsigned)tok * (unsigned)(C3 * 2) + 16u * (unsigned)dlch : 0xFFFFFF00u, 0, 0, 0);
    };
    // fragment addresses (LDS byte addresses for the asm reads).  d: row 4 g + q, bytes 32 mt + 8 p -> chunk 2 mt + (p >> 1), swizzled
    const unsigned lds0 = (unsigned)(size_t)(wg3_lds_void*)smem;
    unsigned a_lane[6], b_lane[2];
#pragma unroll
    for (int mt = 0; mt < 6; ++mt)
        a_lane[mt] = lds0 + (unsigned)(WG3_A2 + (4 * g + q) * 256 + (((2 * mt + (p >> 1)) ^ (2 * ((4 * g + q) & 7))) << 4) + (p & 1) * 8);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) b_lane[nt] = lds0 + (unsigned)(dx * 4096 + off128(4 * g + q, 2 * (2 * nh + nt) + (p >> 1)) + (p & 1) * 8);

    if (s_begin < s_end) issue(s_begin, 0);
    if (s_begin + 1 < s_end) issue(s_begin + 1, 1);
    int stage = 0;
    for (int st = s_begin; st < s_end; ++st) {
        // step st has landed for this wave once only the younger step's three instructions are outstanding; the barrier says so for every
        // wave, and that every wave is done reading step st - 1, whose stage the next issue overwrites
        if (st + 1 < s_end) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (st + 2 < s_end) issue(st + 2, stage == 0 ? 2 : stage - 1);
        const unsigned so = (unsigned)(stage * WG3_STAGE);
        u32x2 al[6], ah[6], bl[2], bh[2];
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) {
            wg3_tr<0>(al[mt], a_lane[mt] + so);
            wg3_tr<16 * 256>(ah[mt], a_lane[mt] + so);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            wg3_tr<0>(bl[nt], b_lane[nt] + so);
            wg3_tr<2048>(bh[nt], b_lane[nt] + so);
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(al[0]), "+v"(ah[0]), "+v"(al[1]), "+v"(ah[1]), "+v"(al[2]), "+v"(ah[2]), "+v"(al[3]), "+v"(ah[3]), "+v"(al[4]),
                       "+v"(ah[4]), "+v"(al[5]), "+v"(ah[5]), "+v"(bl[0]), "+v"(bh[0]), "+v"(bl[1]), "+v"(bh[1])
                     :
                     : "memory");
        bf16x8 a[6];
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) a[mt] = __builtin_bit_cast(bf16x8, u32x4{al[mt][0], al[mt][1], ah[mt][0], ah[mt][1]});
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            // BatchNorm + ReLU on the 8 tokens of this lane's channel (the arithmetic of bn_relu8)
            const u32x4 raw = {bl[nt][0], bl[nt][1], bh[nt][0], bh[nt][1]};
            u32x4 o;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const float lo = fmaxf(__builtin_fmaf(bf_lo(raw[d]), scl[nt], shl[nt]), 0.f);
                const float hi = fmaxf(__builtin_fmaf(bf_hi(raw[d]), scl[nt], shl[nt]), 0.f);
                o[d] = pack_bf16(lo, hi);
            }
            const bf16x8 b = __builtin_bit_cast(bf16x8, o);
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) acc[nt][mt] = mfma32(a[mt], b, acc[nt][mt]);
        }
        stage = stage == WG3_ST - 1 ? 0 : stage + 1;
    }
    float* out = partial + ((size_t)blockIdx.x * gridDim.y + dy) * WG3_OUT + wave * WG3_TILES * 256;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 6; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) out[((nt * 6 + mt) * 4 + e) * 64 + lane] = acc[nt][mt][e];
}

// ---------------------------------------------------------------------------------------------
// Small parameter kernels (one launch each instead of dozens of framework element-wise launches per step)
// ---------------------------------------------------------------------------------------------
// weights f32 -> the packed bf16 operand layouts of the kernels above
__global__ void stem_pack_weights_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                         const float* __restrict__ w3, unsigned short* __restrict__ w1p,
                                         unsigned short* __restrict__ w2p, unsigned short* __restrict__ w2t,
                                         unsigned short* __restrict__ w3p, unsigned short* __restrict__ w3t) {
    constexpr int N1 = C1 * 48, N2 = 9 * C2 * C1, N3 = 16 * C3 * C2;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N1 + 2 * N2 + 2 * N3; i += gridDim.x * blockDim.x) {
        if (i < N1) {                                   // w1p[o][tap][ch] <- w1[o][ch][tap]
            const int o = i / 48, r = i - 48 * o, tap = r >> 2, ch = r & 3;
            w1p[i] = (tap < 9 && ch < 3) ? f32_to_bf16_bits(w1[(o * 3 + ch) * 9 + tap]) : (unsigned short)0;
        } else if (i < N1 + N2) {                       // w2p[tap][o][in] <- w2[o][in][tap]
            const int j = i - N1, tap = j / (C2 * C1), r = j - tap * C2 * C1, o = r / C1, in = r - o * C1;
            w2p[j] = f32_to_bf16_bits(w2[(o * C1 + in) * 9 + tap]);
        } else if (i < N1 + 2 * N2) {                   // w2t[tap][in][o]
            const int j = i - N1 - N2, tap = j / (C2 * C1), r = j - tap * C2 * C1, in = r / C2, o = r - in * C2;
            w2t[j] = f32_to_bf16_bits(w2[(o * C1 + in) * 9 + tap]);
        } else if (i < N1 + 2 * N2 + N3) {              // w3p[tap][o][in] <- w3[o][in][tap]
            const int j = i - N1 - 2 * N2, tap = j / (C3 * C2), r = j - tap * C3 * C2, o = r / C2, in = r - o * C2;
            w3p[j] = f32_to_bf16_bits(w3[(o * C2 + in) * 16 + tap]);
        } else {                                        // w3t[tap][in][o]
            const int j = i - N1 - 2 * N2 - N3, tap = j / (C3 * C2), r = j - tap * C3 * C2, in = r / C3, o = r - in * C3;
            w3t[j] = f32_to_bf16_bits(w3[(o * C2 + in) * 16 + tap]);
        }
    }
}

// BatchNorm folding: statistics of a bias-free convolution output -> prm[4][C] = scale, shift, rstd, -mean rstd;
// training: batch statistics + nn.BatchNorm2d running-statistics update (the conv bias only shifts the tracked mean)
__global__ void stem_bn_fold_kernel(const float* __restrict__ sum, const float* __restrict__ sumsq, double count,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ conv_bias, float eps, float momentum, int training,
                                    float* __restrict__ running_mean, float* __restrict__ running_var, int C,
                                    float* __restrict__ prm, long long* __restrict__ num_batches_tracked) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;       // nn.BatchNorm2d's counter (one launch less per BatchNorm and step)
    const float cb = conv_bias ? conv_bias[c] : 0.f;
    float mean, var;
    if (training) {
        const double m = (double)sum[c] / count;
        double v = (double)sumsq[c] / count - m * m;
        v = v > 0.0 ? v : 0.0;
        mean = (float)m;
        var = (float)v;
        if (running_mean) {
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (mean + cb);
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(v * (count / (count > 1.0 ? count - 1.0 : 1.0)));
        }
    } else {
        mean = running_mean[c] - cb;
        var = running_var[c];
    }
    const float rstd = 1.0f / sqrtf(var + eps);
    const float sc = gamma[c] * rstd;
    prm[c] = sc;
    prm[C + c] = beta[c] - mean * sc;
    prm[2 * C + c] = rstd;
    prm[3 * C + c] = -mean * rstd;
}

// BN2 backward coefficients: sums[2][64] = (sum g, sum g yhat) -> prm5[5][64] = scale, shift, k1, P, Q for conv3_bwd_data
__global__ void stem_bn2_coef_kernel(const float* __restrict__ sums, const float* __restrict__ prm, float count, int training,
                                     float* __restrict__ prm5) {
    const int c = threadIdx.x;
    if (c >= C2) return;
    const float sc = prm[c], sh = prm[C2 + c], a = prm[2 * C2 + c], b = prm[3 * C2 + c];
    float P = 0.f, Q = 0.f;
    if (training) {
        const float m1 = sums[c] / count, m2 = sums[C2 + c] / count;
        P = sc * a * m2;
        Q = sc * (m1 + b * m2);
    }
    prm5[c] = sc;
    prm5[C2 + c] = sh;
    prm5[2 * C2 + c] = sc;
    prm5[3 * C2 + c] = P;
    prm5[4 * C2 + c] = Q;
}

// conv1 weight gradient from the correlations: dW1[o][ch][tap] = k1 (G - m1 X1 - m2 Y)[o][tap*4+ch],
// X1 = XX[ones], Y = rstd (W1 XX - mean X1);  out5 = (sum g1, sum g1 yhat1, G[32][48]);  prm1 = scale, shift, rstd, -mean rstd
__global__ void stem_conv1_wgrad_kernel(const float* __restrict__ out5, const float* __restrict__ xx,
                                        const unsigned short* __restrict__ w1p, const float* __restrict__ prm1, float count,
                                        int training, float* __restrict__ dw1, float* __restrict__ db1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // over 32 * 27
    if (i >= C1 * 27) return;
    const int o = i / 27, r = i - 27 * o, ch = r / 9, tap = r - 9 * ch;
    const int k = tap * 4 + ch;
    const float sc = prm1[o], rstd = prm1[2 * C1 + o], nmr = prm1[3 * C1 + o];     // nmr = -mean rstd
    const float G = out5[2 * C1 + o * 48 + k];
    float v = G;
    if (training) {
        const float m1 = out5[o] / count, m2 = out5[C1 + o] / count;
        const float X1 = xx[(4 * 4 + 3) * 48 + k];
        float syx = 0.f;
        for (int kk = 0; kk < 48; ++kk) syx = __builtin_fmaf(bf16_bits_to_f32(w1p[o * 48 + kk]), xx[kk * 48 + k], syx);
        const float Y = rstd * syx + nmr * X1;
        v = G - m1 * X1 - m2 * Y;
    }
    dw1[i] = sc * v;
    if (r == 0 && db1) db1[o] = training ? 0.f : sc * out5[o];
}

// out[perm[c]] = sum_r part[r][c]: the column sum of the weight-gradient partials, written in parameter layout
__global__ void colsum_perm_kernel(const float* __restrict__ part, int R, int N, const int* __restrict__ perm,
                                   float* __restrict__ out) {
    __shared__ float red[64][16];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
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
        out[perm[c]] = s;
    }
}

namespace t8 {
#define STEM_TH 8
#include "pswin_stem_tiles.inc"
#undef STEM_TH
}  // namespace t8
namespace t16 {
#define STEM_TH 16
#include "pswin_stem_tiles.inc"
#undef STEM_TH
}  // namespace t16

inline int grid_for(int ntiles, int per_cu) { return ntiles < 256 * per_cu ? ntiles : 256 * per_cu; }

}  // namespace

extern "C" {

int pswin_stem_pack_input(const float* x, int B, int H, int W, void* x4, void* stream) {
    PSWIN_CHECK_ARG(x && x4 && B > 0 && H > 0 && W > 0);
    const long long npix = (long long)H * W, total = npix * B;
    hipLaunchKernelGGL(stem_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, npix,
                       total, reinterpret_cast<u64*>(x4));
    PSWIN_LAUNCH_RET();
}

int pswin_stem_workspace(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return PSWIN_ERR_ARG;
    // the largest partial set of any stem kernel: statistics pass (per wave), conv3 data-gradient sums (per wave of
    // every 128-token workgroup), conv3 / conv2 weight gradients (per workgroup)
    const long long M = (long long)B * (H / 4) * (W / 4);
    long long n = 512ll * t8::NW * t8::PART1;   // statistics pass: 512 workgroups of t8::NW waves
    const long long f2 = 256ll * t16::NW * t16::PART2;
    n = n > f2 ? n : f2;
    const long long a = ((M + TOK_WG - 1) / TOK_WG) * TNW * PART3, b = 128ll * 4 * WG3_OUT, c2 = 512ll * t8::WG2_OUT;
    n = n > a ? n : a;
    n = n > b ? n : b;
    n = n > c2 ? n : c2;
    return n > 0x7fffffffll ? PSWIN_ERR_ARG : (int)n;
}

int pswin_stem_conv1_stats(const void* x4, const void* w1p, int B, int H, int W, int want_xx, float* sums,
                           float* workspace, void* stream) {
    PSWIN_CHECK_ARG(x4 && w1p && sums && workspace && B > 0 && H > 0 && W > 0);
    using namespace t8;
    const int nty = (H + TH - 1) / TH, ntx = (W + TW - 1) / TW, ntiles = B * nty * ntx;
    const int grid = grid_for(ntiles, 2);
    hipLaunchKernelGGL(stem_stats1_kernel, dim3(grid), dim3(WG), 0, (hipStream_t)stream, reinterpret_cast<const u64*>(x4), w1p,
                       H, W, nty, ntx, ntiles, want_xx, workspace);
    launch_colsum(workspace, grid * NW, PART1, sums, (hipStream_t)stream);     // fixed-order sum over the per-wave rows
    PSWIN_LAUNCH_RET();
}

int pswin_stem_conv2_fwd(const void* x4, const void* w1p, const float* scale1, const float* shift1, const void* w2p, int B,
                         int H, int W, void* y2, float* sums2, float* workspace, void* stream) {
    PSWIN_CHECK_ARG(x4 && w1p && scale1 && shift1 && w2p && y2 && B > 0 && H > 0 && W > 0);
    PSWIN_CHECK_ARG((long long)H * W * 128 < 0xFFFFFF00ll);
    PSWIN_CHECK_ARG(!sums2 || workspace);
    using namespace t16;
    const int nty = (H + TH - 1) / TH, ntx = (W + TW - 1) / TW, ntiles = B * nty * ntx;
    const int grid = grid_for(ntiles, 1);
    hipLaunchKernelGGL(stem_conv2_fwd_kernel, dim3(grid), dim3(WG), 0, (hipStream_t)stream, reinterpret_cast<const u64*>(x4),
                       w1p, scale1, shift1, w2p, H, W, nty, ntx, ntiles, y2, sums2 ? workspace : nullptr);
    if (sums2) launch_colsum(workspace, grid * NW, PART2, sums2, (hipStream_t)stream);
    PSWIN_LAUNCH_RET();
}

int pswin_stem_conv3_fwd(const void* y2, const float* scale2, const float* shift2, const void* w3p, const float* bias3,
                         int B, int H, int W, void* tokens, void* stream) {
    PSWIN_CHECK_ARG(y2 && scale2 && shift2 && w3p && bias3 && tokens && B > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0);
    const long long M = (long long)B * (H / 4) * (W / 4);
    constexpr int TT = PSWIN_STEM_CONV3_TT;
    const unsigned grid = (unsigned)((M + TOK_WG * TT - 1) / (TOK_WG * TT));
    hipLaunchKernelGGL(stem_conv3_fwd_kernel<TT>, dim3(grid), dim3(TWG), 0, (hipStream_t)stream, y2, scale2, shift2, w3p, bias3, H,
                       W, M, (long long)B * H * W * 128, tokens);
    PSWIN_LAUNCH_RET();
}

int pswin_stem_conv3_bwd_stats(const void* dtok, const void* y2, const float* prm, const void* w3t, int B, int H, int W,
                               float* sums, float* workspace, void* stream) {
    PSWIN_CHECK_ARG(dtok && y2 && prm && w3t && sums && workspace && B > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0);
    const long long M = (long long)B * (H / 4) * (W / 4);
    const long long groups = (M + TOK_WG - 1) / TOK_WG;
    const unsigned grid = (unsigned)(groups < 256 ? groups : 256);          // persistent: one workgroup per CU
    PSWIN_CHECK_ARG((long long)grid * TNW * PART3 <= (long long)pswin_stem_workspace(B, H, W) && M * (long long)(C3 * 2) < 0xFFFFFF00ll);
    static std::atomic<unsigned long long> configured{0};
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&stem_conv3_bwd_kernel<false>), W3H_BYTES, configured)) return rc;
    hipLaunchKernelGGL(stem_conv3_bwd_kernel<false>, dim3(grid), dim3(TWG), W3H_BYTES, (hipStream_t)stream, dtok, y2, prm, w3t, H, W, M,
                       (long long)B * H * W * 128, nullptr, workspace);
    launch_colsum(workspace, (int)grid * TNW, PART3, sums, (hipStream_t)stream);
    PSWIN_LAUNCH_RET();
}

int pswin_stem_conv3_bwd_data(const void* dtok, const void* y2, const float* prm, const void* w3t, int B, int H, int W,
                              void* dy2, void* stream) {
    PSWIN_CHECK_ARG(dtok && y2 && prm && w3t && dy2 && B > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0);
    const long long M = (long long)B * (H / 4) * (W / 4);
    const long long groups = (M + TOK_WG - 1) / TOK_WG;
    const unsigned grid = (unsigned)(groups < 256 ? groups : 256);
    PSWIN_CHECK_ARG(M * (long long)(C3 * 2) < 0xFFFFFF00ll);
    static std::atomic<unsigned long long> configured{0};
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&stem_conv3_bwd_kernel<true>), W3H_BYTES, configured)) return rc;
    hipLaunchKernelGGL(stem_conv3_bwd_kernel<true>, dim3(grid), dim3(TWG), W3H_BYTES, (hipStream_t)stream, dtok, y2, prm, w3t, H, W, M,
                       (long long)B * H * W * 128, dy2, nullptr);
    PSWIN_LAUNCH_RET();
}

int pswin_stem_conv3_wgrad(const void* dtok, const void* y2, const float* scale2, const float* shift2, int B, int H, int W,
                           const int32_t* perm, float* dw3, float* workspace, void* stream) {
    PSWIN_CHECK_ARG(dtok && y2 && scale2 && shift2 && dw3 && workspace && B > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0);
    const long long M = (long long)B * (H / 4) * (W / 4);
    PSWIN_CHECK_ARG(M < 0x7fffffffll - 64);
    const int nsteps = (int)((M + 31) / 32);
    int splits = nsteps < 128 ? nsteps : 128;        // workgroups per tap row: 512 in all = two per CU (126 registers, 47 KB of LDS each) -- with one, a
                                                     // CU has a single 22 KB step in flight and the kernel runs at 2.8 TB/s (bytes in flight / latency)
    const int per = (nsteps + splits - 1) / splits;
    splits = (nsteps + per - 1) / per;
    PSWIN_CHECK_ARG((long long)splits * 4 * WG3_OUT <= (long long)pswin_stem_workspace(B, H, W));
    PSWIN_CHECK_ARG((long long)B * H * W * 128 < 0xFFFFFF00ll && M * (long long)(C3 * 2) < 0xFFFFFF00ll);
    {
        static std::atomic<unsigned long long> configured{0};
        if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&stem_conv3_wgrad_kernel), WG3_LDS, configured)) return rc;
    }
    hipLaunchKernelGGL(stem_conv3_wgrad_kernel, dim3(splits, 4), dim3(TWG), WG3_LDS, (hipStream_t)stream, dtok, y2, scale2, shift2, H,
                       W, (int)M, per, (long long)B * H * W * 128, workspace);
    if (perm)
        hipLaunchKernelGGL(colsum_perm_kernel, dim3((4 * WG3_OUT + 15) / 16), dim3(1024), 0, (hipStream_t)stream, workspace, splits,
                           4 * WG3_OUT, perm, dw3);
    else
        launch_colsum(workspace, splits, 4 * WG3_OUT, dw3, (hipStream_t)stream);
    PSWIN_LAUNCH_RET();
}

int pswin_stem_conv2_wgrad(const void* x4, const void* w1p, const float* scale1, const float* shift1, const void* dy2, int B,
                           int H, int W, const int32_t* perm, float* dw2, float* workspace, void* stream) {
    PSWIN_CHECK_ARG(x4 && w1p && scale1 && shift1 && dy2 && dw2 && workspace && B > 0 && H > 0 && W > 0);
    using namespace t8;
    const int nty = (H + TH - 1) / TH, ntx = (W + TW - 1) / TW, ntiles = B * nty * ntx;
    const int grid = grid_for(ntiles, 2);
    hipLaunchKernelGGL(stem_conv2_wgrad_kernel, dim3(grid), dim3(WG), 0, (hipStream_t)stream, reinterpret_cast<const u64*>(x4),
                       w1p, scale1, shift1, dy2, H, W, nty, ntx, ntiles, workspace);
    if (perm)
        hipLaunchKernelGGL(colsum_perm_kernel, dim3((WG2_OUT + 15) / 16), dim3(1024), 0, (hipStream_t)stream, workspace, grid, WG2_OUT,
                           perm, dw2);
    else
        launch_colsum(workspace, grid, WG2_OUT, dw2, (hipStream_t)stream);
    PSWIN_LAUNCH_RET();
}

int pswin_stem_conv2_bwd(const void* x4, const void* w1p, const float* prm, const void* dy2, const void* w2t, int B, int H,
                         int W, float* out, float* workspace, void* stream) {
    PSWIN_CHECK_ARG(x4 && w1p && prm && dy2 && w2t && out && workspace && B > 0 && H > 0 && W > 0);
    using namespace t16;
    const int nty = (H + TH - 1) / TH, ntx = (W + TW - 1) / TW, ntiles = B * nty * ntx;
    const int grid = grid_for(ntiles, 1);
    hipLaunchKernelGGL(stem_conv2_bwd_kernel, dim3(grid), dim3(WG), 0, (hipStream_t)stream, reinterpret_cast<const u64*>(x4),
                       w1p, prm, dy2, w2t, H, W, nty, ntx, ntiles, workspace);
    launch_colsum(workspace, grid * NW, PART5, out, (hipStream_t)stream);
    PSWIN_LAUNCH_RET();
}

int pswin_stem_pack_weights(const float* w1, const float* w2, const float* w3, void* w1p, void* w2p, void* w2t, void* w3p,
                            void* w3t, void* stream) {
    PSWIN_CHECK_ARG(w1 && w2 && w3 && w1p && w2p && w2t && w3p && w3t);
    hipLaunchKernelGGL(stem_pack_weights_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, w1, w2, w3,
                       (unsigned short*)w1p, (unsigned short*)w2p, (unsigned short*)w2t, (unsigned short*)w3p, (unsigned short*)w3t);
    PSWIN_LAUNCH_RET();
}

int pswin_stem_bn_fold(const float* sum, const float* sumsq, double count, const float* gamma, const float* beta,
                       const float* conv_bias, float eps, float momentum, int training, float* running_mean,
                       float* running_var, int C, float* prm, long long* num_batches_tracked, void* stream) {
    PSWIN_CHECK_ARG(gamma && beta && prm && C > 0 && C <= 1024 && count > 0);
    PSWIN_CHECK_ARG(training ? (sum && sumsq) : (running_mean && running_var));
    hipLaunchKernelGGL(stem_bn_fold_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, sum, sumsq, count, gamma,
                       beta, conv_bias, eps, momentum, training, running_mean, running_var, C, prm, num_batches_tracked);
    PSWIN_LAUNCH_RET();
}

int pswin_stem_bn2_coefs(const float* sums, const float* prm, double count, int training, float* prm5, void* stream) {
    PSWIN_CHECK_ARG(sums && prm && prm5 && count > 0);
    hipLaunchKernelGGL(stem_bn2_coef_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, prm, (float)count, training, prm5);
    PSWIN_LAUNCH_RET();
}

int pswin_stem_conv1_wgrad(const float* out5, const float* xx, const void* w1p, const float* prm1, double count, int training,
                           float* dw1, float* db1, void* stream) {
    PSWIN_CHECK_ARG(out5 && w1p && prm1 && dw1 && count > 0 && (!training || xx));
    hipLaunchKernelGGL(stem_conv1_wgrad_kernel, dim3((C1 * 27 + 63) / 64), dim3(64), 0, (hipStream_t)stream, out5, xx,
                       (const unsigned short*)w1p, prm1, (float)count, training, dw1, db1);
    PSWIN_LAUNCH_RET();
}

}  // extern "C"
