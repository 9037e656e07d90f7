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
constexpr int TH = 16, TW = 32;              // output tile of the 3x3 kernels
constexpr int WG = 512;                      // 8 waves
constexpr int NW = WG / 64;
constexpr int XR = TH + 4, XC = TW + 4;      // input tile with a halo of 2
constexpr int AR = TH + 2, AC = TW + 2;      // a1 / dy2 tile with a halo of 1
constexpr int A_PIX = AR * AC;               // 612
constexpr int A_GROUPS = (A_PIX + 15) / 16;  // 39 groups of 16 pixels
constexpr int XS_BYTES = XR * XC * 8;        // 5760
constexpr int ZSLOT = XS_BYTES;              // 16 zero bytes behind the input tile
constexpr int XS_TOTAL = XS_BYTES + 16;

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

__device__ inline unsigned pack_bf16(float lo, float hi) {
    return (unsigned)f32_to_bf16_bits(lo) | ((unsigned)f32_to_bf16_bits(hi) << 16);
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

struct Tile {
    int b, y0, x0;
};
__device__ inline Tile tile_of(int t, int nty, int ntx) {
    Tile r;
    r.b = t / (nty * ntx);
    const int rem = t - r.b * nty * ntx;
    const int ty = rem / ntx;
    r.y0 = ty * TH;
    r.x0 = (rem - ty * ntx) * TW;
    return r;
}

// input tile with halo 2 -> LDS (8 bytes per pixel, zero outside the image), plus the zero slot
__device__ inline void load_xs(const u64* __restrict__ x4, const Tile& t, int H, int W, char* xs) {
    for (int i = threadIdx.x; i < XR * XC; i += WG) {
        const int r = i / XC, cc = i - r * XC;
        const int gy = t.y0 - 2 + r, gx = t.x0 - 2 + cc;
        u64 v = 0;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = x4[((size_t)t.b * H + gy) * W + gx];
        reinterpret_cast<u64*>(xs)[i] = v;
    }
    if (threadIdx.x < 2) reinterpret_cast<u64*>(xs + ZSLOT)[threadIdx.x] = 0;
}

// conv1 weights as MFMA A operands: rows = output channel 16 nt + c, k = the 4 channel slots of tap 4 s + g
struct W1Frags {
    s16x4 a[2][3];
};
__device__ inline W1Frags load_w1(const void* w1p, int c, int g) {
    W1Frags f;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int s = 0; s < 3; ++s)
            f.a[nt][s] = *reinterpret_cast<const s16x4*>(reinterpret_cast<const char*>(w1p) + ((16 * nt + c) * 12 + 4 * s + g) * 8);
    return f;
}

// conv1 for the 16 pixels (py, px0 + c) of a region whose pixel (0, 0) sits at xs[oy][ox] minus one row/column of taps:
// acc[nt][e] = y1[channel 16 nt + 4 g + e][pixel c]
__device__ inline void conv1_group(const char* xs, const W1Frags& w, int py, int px, int oy, int ox, int g, f32x4 (&acc)[2]) {
    acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        int tap = 4 * s + g;
        tap = tap < 9 ? tap : 0;                  // slots 9..11: zero weights, any finite operand
        const int ty = tap / 3, tx = tap - 3 * ty;
        const s16x4 b = *reinterpret_cast<const s16x4*>(xs + ((py + ty + oy) * XC + px + tx + ox) * 8);
        acc[0] = mfma16(w.a[0][s], b, acc[0]);
        acc[1] = mfma16(w.a[1][s], b, acc[1]);
    }
}

// a1 = relu(bn1(conv1 x)) on the tile with halo 1 -> LDS image [612 (+pad) pixels][32 ch] (off64), zero outside the image
__device__ inline void build_a1(const char* xs, const W1Frags& w, const float (&sc)[2][4], const float (&sh)[2][4],
                                const Tile& t, int H, int W, int wave, int c, int g, char* a1s) {
    for (int grp = wave; grp < A_GROUPS; grp += NW) {
        const int P = 16 * grp + c;
        const int Pc = P < A_PIX ? P : A_PIX - 1;
        const int py = Pc / AC, px = Pc - py * AC;
        f32x4 acc[2];
        conv1_group(xs, w, py, px, 0, 0, g, acc);
        const int gy = t.y0 - 1 + py, gx = t.x0 - 1 + px;
        const bool in = (P < A_PIX) && gy >= 0 && gy < H && gx >= 0 && gx < W;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float z = fmaxf(__builtin_fmaf(acc[nt][e], sc[nt][e], sh[nt][e]), 0.f);
                v[e] = in ? z : 0.f;
            }
            u32x2 pk = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
            *reinterpret_cast<u32x2*>(a1s + off64(P, 2 * nt + (g >> 1)) + (g & 1) * 8) = pk;
        }
    }
}

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
// F1: statistics of y1 = conv1(x) (never stored) and the input autocorrelation XX
// ---------------------------------------------------------------------------------------------
constexpr int PART1 = 2 * C1 + 48 * 48;   // per-wave partial row: sum y1 [32], sum y1^2 [32], XX [48][48]

__global__ __launch_bounds__(WG) void stem_stats1_kernel(const u64* __restrict__ x4, const void* __restrict__ w1p, int H,
                                                         int W, int nty, int ntx, int ntiles, int want_xx,
                                                         float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) char xs[XS_TOTAL];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
    const W1Frags w = load_w1(w1p, c, g);
    float s1[2][4], q1[2][4];
    f32x4 xx[3][3];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) s1[nt][e] = q1[nt][e] = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) xx[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const Tile t = tile_of(tile, nty, ntx);
        __syncthreads();
        load_xs(x4, t, H, W, xs);
        __syncthreads();
        // y1 on the tile pixels: group = half a tile row
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int grp = wave * 4 + k;
            const int py = grp >> 1, px = 16 * (grp & 1) + c;
            f32x4 acc[2];
            conv1_group(xs, w, py, px, 1, 1, g, acc);
            const bool in = (t.y0 + py < H) && (t.x0 + px < W);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = in ? acc[nt][e] : 0.f;
                    s1[nt][e] += v;
                    q1[nt][e] = __builtin_fmaf(v, v, q1[nt][e]);
                }
        }
        if (want_xx) {
            // XX += xp^T xp over the valid pixels of tile rows 2 wave, 2 wave + 1 (one 32-pixel contraction step each)
            const int q = c >> 2, p = c & 3;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int y = 2 * wave + k;
                const bool rowin = t.y0 + y < H;
                bf16x8 f[3];
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) {
                    const int tap = 4 * nb + p;
                    const int ty = tap / 3, tx = tap - 3 * ty;
                    const char* lo = tap < 9 ? xs + ((y + ty + 1) * XC + 4 * g + q + tx + 1) * 8 : xs + ZSLOT;
                    const char* hi = tap < 9 ? lo + 16 * 8 : lo;
                    u32x4 raw = __builtin_bit_cast(u32x4, tr_pair(lo, hi));
                    // element r of the lo (hi) half belongs to pixel column 4 g + r (16 + 4 g + r)
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int col = (d >> 1) * 16 + 4 * g + 2 * (d & 1);
                        const bool in0 = rowin && (t.x0 + col < W), in1 = rowin && (t.x0 + col + 1 < W);
                        raw[d] &= (in0 ? 0x0000ffffu : 0u) | (in1 ? 0xffff0000u : 0u);
                    }
                    f[nb] = __builtin_bit_cast(bf16x8, raw);
                }
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) xx[a][b] = mfma32(f[a], f[b], xx[a][b]);
            }
        }
    }
    float* out = partial + ((size_t)blockIdx.x * NW + wave) * PART1;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = row16_sum(s1[nt][e]), b = row16_sum(q1[nt][e]);
            if (c == 0) {
                out[16 * nt + 4 * g + e] = a;
                out[C1 + 16 * nt + 4 * g + e] = b;
            }
        }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) out[2 * C1 + (16 * a + 4 * g + e) * 48 + 16 * b + c] = xx[a][b][e];
}

// ---------------------------------------------------------------------------------------------
// F2: y2 = conv2(relu(bn1(conv1 x))) + per-channel sum / sum of squares of y2
// ---------------------------------------------------------------------------------------------
constexpr int A1S_BYTES = A_GROUPS * 16 * 64;     // 39936
constexpr int W2S_BYTES = 9 * C2 * 64;            // 36864
constexpr int PART2 = 2 * C2;

__global__ __launch_bounds__(WG) void stem_conv2_fwd_kernel(const u64* __restrict__ x4, const void* __restrict__ w1p,
                                                            const float* __restrict__ scale1, const float* __restrict__ shift1,
                                                            const void* __restrict__ w2p, int H, int W, int nty, int ntx,
                                                            int ntiles, void* __restrict__ y2, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) char smem[XS_TOTAL + A1S_BYTES + W2S_BYTES];
    char* xs = smem;
    char* a1s = smem + XS_TOTAL;
    char* w2s = a1s + A1S_BYTES;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
    const W1Frags w = load_w1(w1p, c, g);
    float sc[2][4], sh[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sc[nt][e] = scale1[16 * nt + 4 * g + e];
            sh[nt][e] = shift1[16 * nt + 4 * g + e];
        }
    // conv2 weights [9][64 out][32 in] -> LDS rows of 64 bytes (row = tap * 64 + out)
    for (int i = threadIdx.x; i < 9 * C2 * 4; i += WG) {
        const int row = i >> 2, ch = i & 3;
        *reinterpret_cast<u32x4*>(w2s + off64(row, ch)) = reinterpret_cast<const u32x4*>(w2p)[i];
    }
    float s2[4][4], q2[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) s2[mt][e] = q2[mt][e] = 0.f;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const Tile t = tile_of(tile, nty, ntx);
        __syncthreads();                       // previous tile's readers are done
        load_xs(x4, t, H, W, xs);
        __syncthreads();
        build_a1(xs, w, sc, sh, t, H, W, wave, c, g, a1s);
        __syncthreads();
        // wave: tile rows 2 wave, 2 wave + 1, both 16-pixel halves; acc[pt][mt][e] = y2[channel 16 mt + 4 g + e][pixel c]
        f32x4 acc[4][4];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[pt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ty = tap / 3, tx = tap - 3 * ty;
            bf16x8 a[4], b[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const bf16x8*>(w2s + off64(tap * C2 + 16 * mt + c, g));
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                const int r = 2 * wave + (pt >> 1), col = 16 * (pt & 1) + c;
                b[pt] = *reinterpret_cast<const bf16x8*>(a1s + off64((r + ty) * AC + col + tx, g));
            }
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[pt][mt] = mfma32(a[mt], b[pt], acc[pt][mt]);
        }
        // epilogue: statistics + bf16 store (8 consecutive channels per lane after the group exchange)
        const rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(y2) + (size_t)t.b * H * W * 128, 0,
                                                            (int)((size_t)H * W * 128), 0x00020000);
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const int gy = t.y0 + 2 * wave + (pt >> 1), gx = t.x0 + 16 * (pt & 1) + c;
            const bool in = gy < H && gx < W;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = in ? acc[pt][mt][e] : 0.f;
                    s2[mt][e] += v;
                    q2[mt][e] = __builtin_fmaf(v, v, q2[mt][e]);
                }
            const unsigned poff = in ? (unsigned)(gy * W + gx) * 128u + (unsigned)row8_d0(g) * 2u : 0xFFFFFF00u;
            __builtin_amdgcn_raw_buffer_store_b128(pack_row8(g, acc[pt][0], acc[pt][1]), ys, poff, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(pack_row8(g, acc[pt][2], acc[pt][3]), ys, poff + 64u, 0, 0);
        }
    }
    if (partial) {
        float* out = partial + ((size_t)blockIdx.x * NW + wave) * PART2;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = row16_sum(s2[mt][e]), b = row16_sum(q2[mt][e]);
                if (c == 0) {
                    out[16 * mt + 4 * g + e] = a;
                    out[C2 + 16 * mt + 4 * g + e] = b;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// F3: tokens = conv3(relu(bn2(y2))) + bias     (4x4 stride 4: one GEMM row per token, K = 16 taps x 64 channels)
// ---------------------------------------------------------------------------------------------
constexpr int W3S_BYTES = C3 * 128;          // one tap slice [96 out][64 in]
constexpr int TOK_WG = NW * 16;              // 128 tokens per workgroup

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

__global__ __launch_bounds__(WG) void stem_conv3_fwd_kernel(const void* __restrict__ y2, const float* __restrict__ scale2,
                                                            const float* __restrict__ shift2, const void* __restrict__ w3p,
                                                            const float* __restrict__ bias3, int H, int W, long long M,
                                                            long long y2_bytes, void* __restrict__ tok_out) {
    __shared__ __attribute__((aligned(16))) char w3s[2][W3S_BYTES];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
    const int Hh = H / 4, Wh = W / 4;
    const long long tok_wg = (long long)blockIdx.x * TOK_WG;
    long long tok = tok_wg + wave * 16 + c;
    const bool valid = tok < M;
    if (!valid) tok = M - 1;
    const long long base_pix = token_pix(tok_wg, Hh, Wh, H, W);                 // uniform
    const long long rem_bytes = y2_bytes - base_pix * 128;
    const rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(y2)) + base_pix * 128, 0,
                                                        (int)(rem_bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : rem_bytes), 0x00020000);
    const unsigned voff = (unsigned)((token_pix(tok, Hh, Wh, H, W) - base_pix) * 128) + 16u * g;
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
    const int ch0 = threadIdx.x, ch1 = threadIdx.x + WG;
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
    u32x4 yb[2], yn[2];
    yb[0] = __builtin_amdgcn_raw_buffer_load_b128(ys, voff, 0, 0);
    yb[1] = __builtin_amdgcn_raw_buffer_load_b128(ys, voff + 64u, 0, 0);
    f32x4 acc[6];
#pragma unroll
    for (int mt = 0; mt < 6; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int tap = 0; tap < 16; ++tap) {
        const int buf = tap & 1;
        if (tap + 1 < 16) {
            stage_load(tap + 1, r0, r1);
            const int nt = tap + 1;
            const int soff = ((nt >> 2) * W + (nt & 3)) * 128;
            yn[0] = __builtin_amdgcn_raw_buffer_load_b128(ys, voff, soff, 0);
            yn[1] = __builtin_amdgcn_raw_buffer_load_b128(ys, voff + 64u, soff, 0);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8 b = bn_relu8(yb[s], sc[s], sh[s]);
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(w3s[buf] + off128(16 * mt + c, 4 * s + g));
                acc[mt] = mfma32(a, b, acc[mt]);
            }
        }
        if (tap + 1 < 16) {
            stage_store(buf ^ 1, r0, r1);
            yb[0] = yn[0];
            yb[1] = yn[1];
        }
        __syncthreads();
    }
    // + bias, bf16, 8 consecutive channels per lane
#pragma unroll
    for (int mt = 0; mt < 6; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mt][e] += bias3[16 * mt + 4 * g + e];
    char* dst = reinterpret_cast<char*>(tok_out) + tok * (C3 * 2) + row8_d0(g) * 2;
#pragma unroll
    for (int pr = 0; pr < 3; ++pr) {
        const u32x4 v = pack_row8(g, acc[2 * pr], acc[2 * pr + 1]);
        if (valid) *reinterpret_cast<u32x4*>(dst + pr * 64) = v;
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

template <bool APPLY>
__global__ __launch_bounds__(WG) void stem_conv3_bwd_kernel(const void* __restrict__ dtok, const void* __restrict__ y2,
                                                            const float* __restrict__ prm, const void* __restrict__ w3t,
                                                            int H, int W, long long M, long long y2_bytes,
                                                            void* __restrict__ dy2, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) char w3s[2][W3T_BYTES];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
    const int Hh = H / 4, Wh = W / 4;
    const long long tok_wg = (long long)blockIdx.x * TOK_WG;
    long long tok = tok_wg + wave * 16 + c;
    const bool valid = tok < M;
    if (!valid) tok = M - 1;
    const long long base_pix = token_pix(tok_wg, Hh, Wh, H, W);
    const long long rem_bytes = y2_bytes - base_pix * 128;
    const int nrec = (int)(rem_bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : rem_bytes);
    const rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(y2)) + base_pix * 128, 0, nrec, 0x00020000);
    const rsrc_t ds = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(dy2) + base_pix * 128, 0, APPLY ? nrec : 0, 0x00020000);
    const int d0 = row8_d0(g);
    const unsigned voff = (unsigned)((token_pix(tok, Hh, Wh, H, W) - base_pix) * 128) + 2u * d0;
    // per-lane channels: 32 h + d0 + j
    float p0[2][8], p1[2][8], p2[2][8], p3[2][8], p4[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ch = 32 * h + d0 + j;
            p0[h][j] = prm[ch];
            p1[h][j] = prm[C2 + ch];
            p2[h][j] = prm[2 * C2 + ch];
            p3[h][j] = prm[3 * C2 + ch];
            p4[h][j] = APPLY ? prm[4 * C2 + ch] : 0.f;
        }
    // d tokens as B operands: k = output channel 32 s + 8 g ..
    bf16x8 bt[3];
    {
        const char* src = reinterpret_cast<const char*>(dtok) + tok * (C3 * 2) + 16 * g;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            u32x4 raw = *reinterpret_cast<const u32x4*>(src + 64 * s);
            if (!valid) raw = u32x4{0u, 0u, 0u, 0u};
            bt[s] = __builtin_bit_cast(bf16x8, raw);
        }
    }
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(w3t);
    const int ch0 = threadIdx.x, ch1 = threadIdx.x + WG;
    auto stage_load = [&](int tap, u32x4& r0, u32x4& r1) {
        r0 = wsrc[tap * 768 + ch0];
        if (ch1 < 768) r1 = wsrc[tap * 768 + ch1];
    };
    auto stage_store = [&](int buf, const u32x4& r0, const u32x4& r1) {
        *reinterpret_cast<u32x4*>(w3s[buf] + (ch0 / 12) * W3T_LD + (ch0 % 12) * 16) = r0;
        if (ch1 < 768) *reinterpret_cast<u32x4*>(w3s[buf] + (ch1 / 12) * W3T_LD + (ch1 % 12) * 16) = r1;
    };
    u32x4 r0, r1 = {0u, 0u, 0u, 0u};
    stage_load(0, r0, r1);
    stage_store(0, r0, r1);
    u32x4 yb[2], yn[2];
    yb[0] = __builtin_amdgcn_raw_buffer_load_b128(ys, voff, 0, 0);
    yb[1] = __builtin_amdgcn_raw_buffer_load_b128(ys, voff + 64u, 0, 0);
    float sg[2][8], sgy[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 8; ++j) sg[h][j] = sgy[h][j] = 0.f;
    __syncthreads();
    for (int tap = 0; tap < 16; ++tap) {
        const int buf = tap & 1;
        const int soff = ((tap >> 2) * W + (tap & 3)) * 128;
        if (tap + 1 < 16) {
            stage_load(tap + 1, r0, r1);
            const int nt = tap + 1;
            const int so = ((nt >> 2) * W + (nt & 3)) * 128;
            yn[0] = __builtin_amdgcn_raw_buffer_load_b128(ys, voff, so, 0);
            yn[1] = __builtin_amdgcn_raw_buffer_load_b128(ys, voff + 64u, so, 0);
        }
        f32x4 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(w3s[buf] + (16 * mt + c) * W3T_LD + (4 * s + g) * 16);
                acc[mt] = mfma32(a, bt[s], acc[mt]);
            }
        }
        // acc[mt][e] = d a2[channel 16 mt + 4 g + e][token c] -> 8 consecutive channels 32 h + d0 .. per lane
        exchange_row8(acc[0], acc[1]);
        exchange_row8(acc[2], acc[3]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float y[8], v[8];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                y[2 * d] = bf_lo(yb[h][d]);
                y[2 * d + 1] = bf_hi(yb[h][d]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float da = acc[2 * h + (j >> 2)][j & 3];
                const bool on = __builtin_fmaf(y[j], p0[h][j], p1[h][j]) > 0.f;
                const float gg = on ? da : 0.f;
                if constexpr (APPLY) {
                    v[j] = __builtin_fmaf(p2[h][j], gg, -__builtin_fmaf(p3[h][j], y[j], p4[h][j]));
                } else {
                    const float yh = __builtin_fmaf(y[j], p2[h][j], p3[h][j]);
                    sg[h][j] += gg;
                    sgy[h][j] = __builtin_fmaf(gg, yh, sgy[h][j]);
                }
            }
            if constexpr (APPLY) {
                const u32x4 o = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
                if (valid) __builtin_amdgcn_raw_buffer_store_b128(o, ds, voff + 64u * h, soff, 0);
            }
        }
        if (tap + 1 < 16) {
            stage_store(buf ^ 1, r0, r1);
            yb[0] = yn[0];
            yb[1] = yn[1];
        }
        __syncthreads();
    }
    if constexpr (!APPLY) {
        float* out = partial + ((size_t)blockIdx.x * NW + wave) * PART3;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = row16_sum(sg[h][j]), b = row16_sum(sgy[h][j]);
                if (c == 0) {
                    out[32 * h + d0 + j] = a;
                    out[C2 + 32 * h + d0 + j] = b;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// B3: conv3 weight gradient  dW3[out][tap][in] = sum_tokens dtok[token][out] * a2[token, tap][in], a2 = relu(bn2 y2)
// recomputed on load.  Workgroup = one tap row dy, a range of 32-token steps; wave = (tap pair th, step residue kq).
// ---------------------------------------------------------------------------------------------
constexpr int DTT_LD = 224;                         // 96 bf16 + 32 bytes: transposed reads conflict free
constexpr int WG3_WAVE_BYTES = 2 * 32 * 128 + 32 * DTT_LD;     // 15360
constexpr int WG3_TILES = 2 * 4 * 6;                // accumulator tiles per wave
constexpr int WG3_OUT = 2 * WG3_TILES * 256;        // floats per workgroup partial: [th][tile][e][lane]

__global__ __launch_bounds__(WG) void stem_conv3_wgrad_kernel(const void* __restrict__ dtok, const void* __restrict__ y2,
                                                              const float* __restrict__ scale2, const float* __restrict__ shift2,
                                                              int H, int W, long long M, int steps_per_wg,
                                                              float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) char smem[NW * WG3_WAVE_BYTES];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4, q = c >> 2, p = c & 3;
    const int th = wave & 1, kq = wave >> 1;
    const int dy = blockIdx.y;
    const int Hh = H / 4, Wh = W / 4;
    char* a2t = smem + wave * WG3_WAVE_BYTES;         // [2 taps][32 tokens][64 ch] (off128)
    char* dtt = a2t + 2 * 32 * 128;                   // [32 tokens][DTT_LD]
    const int chunk = lane & 7;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = scale2[8 * chunk + j];
        sh[j] = shift2[8 * chunk + j];
    }
    f32x4 acc[2][4][6];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) acc[j][nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long nsteps = (M + 31) / 32;
    const long long s_begin = (long long)blockIdx.x * steps_per_wg;
    long long s_end = s_begin + steps_per_wg;
    if (s_end > nsteps) s_end = nsteps;
    for (long long st = s_begin + kq; st < s_end; st += 4) {
        const long long tok0 = st * 32;
        // a2 of this wave's two taps: lane -> (token (lane >> 3) + 8 i, 16-byte channel chunk lane & 7)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tl = (lane >> 3) + 8 * i;
            const long long tok = tok0 + tl;
            const bool ok = tok < M;
            const long long pix = token_pix(ok ? tok : M - 1, Hh, Wh, H, W) + (long long)dy * W + 2 * th;
            const char* src = reinterpret_cast<const char*>(y2) + pix * 128 + 16 * chunk;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                u32x4 raw = *reinterpret_cast<const u32x4*>(src + 128 * j);
                bf16x8 v = bn_relu8(raw, sc, sh);
                if (!ok) v = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
                *reinterpret_cast<bf16x8*>(a2t + j * 32 * 128 + off128(tl, chunk)) = v;
            }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int idx = lane + 64 * i;
            const int tl = idx / 12, ci = idx - 12 * tl;
            const long long tok = tok0 + tl;
            u32x4 raw = {0u, 0u, 0u, 0u};
            if (tok < M) raw = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(dtok) + tok * (C3 * 2) + 16 * ci);
            *reinterpret_cast<u32x4*>(dtt + tl * DTT_LD + 16 * ci) = raw;
        }
        __builtin_amdgcn_wave_barrier();
        bf16x8 a[6];
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) {
            const char* lo = dtt + (4 * g + q) * DTT_LD + (16 * mt + 4 * p) * 2;
            a[mt] = tr_pair(lo, lo + 16 * DTT_LD);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const char* lo = a2t + j * 32 * 128 + off128(4 * g + q, 2 * nt + (p >> 1)) + (p & 1) * 8;
                const bf16x8 b = tr_pair(lo, lo + 16 * 128);
#pragma unroll
                for (int mt = 0; mt < 6; ++mt) acc[j][nt][mt] = mfma32(a[mt], b, acc[j][nt][mt]);
            }
        __builtin_amdgcn_wave_barrier();
    }
    // sum the 4 kq waves of each tap pair through LDS (fixed order), the last one writes the workgroup partial
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem) + th * WG3_TILES * 256;
    float* out = partial + ((size_t)blockIdx.x * gridDim.y + dy) * WG3_OUT + th * WG3_TILES * 256;
    for (int r = 0; r < 4; ++r) {
        if (kq == r) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 6; ++mt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int idx = (((j * 4 + nt) * 6 + mt) * 4 + e) * 64 + lane;
                            float v = acc[j][nt][mt][e];
                            if (r > 0) v += red[idx];
                            if (r < 3) red[idx] = v; else out[idx] = v;
                        }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// B4: conv2 weight gradient  dW2[out][tap][in] = sum_p dy2[p][out] * a1[p + tap][in], a1 recomputed from the input.
// wave = (output-channel half mh, 4 tile rows kq); one tile row (32 pixels) per contraction step.
// ---------------------------------------------------------------------------------------------
constexpr int DYS_BYTES = TH * TW * 128;             // 65536
constexpr int WG2_TILES = 2 * 9 * 2;                 // accumulator tiles per wave
constexpr int WG2_OUT = 2 * WG2_TILES * 256;

__global__ __launch_bounds__(WG) void stem_conv2_wgrad_kernel(const u64* __restrict__ x4, const void* __restrict__ w1p,
                                                              const float* __restrict__ scale1, const float* __restrict__ shift1,
                                                              const void* __restrict__ dy2, int H, int W, int nty, int ntx,
                                                              int ntiles, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) char smem[XS_TOTAL + A1S_BYTES + DYS_BYTES];
    char* xs = smem;
    char* a1s = smem + XS_TOTAL;
    char* dys = a1s + A1S_BYTES;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4, q = c >> 2, p = c & 3;
    const int mh = wave & 1, kq = wave >> 1;
    const W1Frags w = load_w1(w1p, c, g);
    float sc[2][4], sh[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sc[nt][e] = scale1[16 * nt + 4 * g + e];
            sh[nt][e] = shift1[16 * nt + 4 * g + e];
        }
    f32x4 acc[2][9][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[mi][tap][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const Tile t = tile_of(tile, nty, ntx);
        __syncthreads();
        load_xs(x4, t, H, W, xs);
        for (int i = threadIdx.x; i < TH * TW * 8; i += WG) {
            const int pix = i >> 3, ch = i & 7;
            const int gy = t.y0 + (pix >> 5), gx = t.x0 + (pix & 31);
            u32x4 v = {0u, 0u, 0u, 0u};
            if (gy < H && gx < W) v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(dy2) + (((size_t)t.b * H + gy) * W + gx) * 128 + 16 * ch);
            *reinterpret_cast<u32x4*>(dys + off128(pix, ch)) = v;
        }
        __syncthreads();
        build_a1(xs, w, sc, sh, t, H, W, wave, c, g, a1s);
        __syncthreads();
#pragma unroll 1
        for (int yy = 0; yy < 4; ++yy) {
            const int y = 4 * kq + yy;
            bf16x8 a[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int mt = 2 * mh + mi;
                const char* lo = dys + off128(y * TW + 4 * g + q, 2 * mt + (p >> 1)) + (p & 1) * 8;
                a[mi] = tr_pair(lo, lo + 16 * 128);
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ty = tap / 3, tx = tap - 3 * ty;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const char* lo = a1s + off64((y + ty) * AC + 4 * g + q + tx, 2 * nt + (p >> 1)) + (p & 1) * 8;
                    const bf16x8 b = tr_pair(lo, lo + 16 * 64);
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) acc[mi][tap][nt] = mfma32(a[mi], b, acc[mi][tap][nt]);
                }
            }
        }
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem) + mh * WG2_TILES * 256;
    float* out = partial + (size_t)blockIdx.x * WG2_OUT + mh * WG2_TILES * 256;
    for (int r = 0; r < 4; ++r) {
        if (kq == r) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int idx = (((mi * 9 + tap) * 2 + nt) * 4 + e) * 64 + lane;
                            float v = acc[mi][tap][nt][e];
                            if (r > 0) v += red[idx];
                            if (r < 3) red[idx] = v; else out[idx] = v;
                        }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// B5: conv2 data gradient da1, BN1 backward sums and the conv1 weight-gradient correlation G, g1 never stored:
//   da1[p][in] = sum_tap dy2[p - tap][out] W2[tap][out][in];  g1 = da1 [relu(bn1 y1) > 0] (y1 recomputed);
//   out: sum g1 [32], sum g1 * yhat1 [32], G[32][48] = sum_p g1[p][ch] * xp[p][slot].
// prm: f32 [4][32] = scale1, shift1, a = rstd1, b = -mean1 rstd1.
// ---------------------------------------------------------------------------------------------
constexpr int DYH_BYTES = A_GROUPS * 16 * 128;       // 79872: dy2 tile with halo 1
constexpr int W2T_BYTES = 9 * C1 * 128;              // 36864
constexpr int PART5 = 2 * C1 + C1 * 48;              // 1600

__global__ __launch_bounds__(WG) void stem_conv2_bwd_kernel(const u64* __restrict__ x4, const void* __restrict__ w1p,
                                                            const float* __restrict__ prm, const void* __restrict__ dy2,
                                                            const void* __restrict__ w2t, int H, int W, int nty, int ntx,
                                                            int ntiles, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) char smem[XS_TOTAL + DYH_BYTES + W2T_BYTES];
    char* xs = smem;
    char* dyh = smem + XS_TOTAL;
    char* g1s = dyh;                                  // aliases the dy2 tile once the data gradient is done
    char* w2s = dyh + DYH_BYTES;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4, q = c >> 2, p = c & 3;
    const W1Frags w = load_w1(w1p, c, g);
    float sc[2][4], sh[2][4], pa[2][4], pb[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ch = 16 * nt + 4 * g + e;
            sc[nt][e] = prm[ch];
            sh[nt][e] = prm[C1 + ch];
            pa[nt][e] = prm[2 * C1 + ch];
            pb[nt][e] = prm[3 * C1 + ch];
        }
    for (int i = threadIdx.x; i < 9 * C1 * 8; i += WG) {       // [9][32 in][64 out] -> rows of 128 bytes
        const int row = i >> 3, ch = i & 7;
        *reinterpret_cast<u32x4*>(w2s + off128(row, ch)) = reinterpret_cast<const u32x4*>(w2t)[i];
    }
    float sg[2][4], sgy[2][4];
    f32x4 gacc[2][3];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) sg[nt][e] = sgy[nt][e] = 0.f;
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) gacc[nt][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const Tile t = tile_of(tile, nty, ntx);
        __syncthreads();
        load_xs(x4, t, H, W, xs);
        for (int i = threadIdx.x; i < A_PIX * 8; i += WG) {
            const int P = i >> 3, ch = i & 7;
            const int py = P / AC, px = P - py * AC;
            const int gy = t.y0 - 1 + py, gx = t.x0 - 1 + px;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(dy2) + (((size_t)t.b * H + gy) * W + gx) * 128 + 16 * ch);
            *reinterpret_cast<u32x4*>(dyh + off128(P, ch)) = v;
        }
        __syncthreads();
        f32x4 acc[4][2];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) acc[pt][0] = acc[pt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ty = tap / 3, tx = tap - 3 * ty;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 a[2], b[4];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) a[mt] = *reinterpret_cast<const bf16x8*>(w2s + off128(tap * C1 + 16 * mt + c, 4 * s + g));
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) {
                    const int r = 2 * wave + (pt >> 1), col = 16 * (pt & 1) + c;
                    b[pt] = *reinterpret_cast<const bf16x8*>(dyh + off128((r - ty + 2) * AC + col - tx + 2, 4 * s + g));
                }
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[pt][mt] = mfma32(a[mt], b[pt], acc[pt][mt]);
            }
        }
        // mask by the ReLU of the recomputed a1, BN1 backward sums
        u32x2 gp[4][2];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const int r = 2 * wave + (pt >> 1), col = 16 * (pt & 1) + c;
            f32x4 y1[2];
            conv1_group(xs, w, r, col, 1, 1, g, y1);
            const bool in = (t.y0 + r < H) && (t.x0 + col < W);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool on = in && (__builtin_fmaf(y1[nt][e], sc[nt][e], sh[nt][e]) > 0.f);
                    const float gg = on ? acc[pt][nt][e] : 0.f;
                    const float yh = __builtin_fmaf(y1[nt][e], pa[nt][e], pb[nt][e]);
                    sg[nt][e] += gg;
                    sgy[nt][e] = __builtin_fmaf(gg, yh, sgy[nt][e]);
                    v[e] = gg;
                }
                gp[pt][nt] = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
            }
        }
        __syncthreads();                               // every wave is done with the dy2 tile
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const int pix = (2 * wave + (pt >> 1)) * TW + 16 * (pt & 1) + c;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) *reinterpret_cast<u32x2*>(g1s + off64(pix, 2 * nt + (g >> 1)) + (g & 1) * 8) = gp[pt][nt];
        }
        __builtin_amdgcn_wave_barrier();               // a wave reads back only the two tile rows it wrote
        // G += g1^T xp over tile rows 2 wave, 2 wave + 1
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int y = 2 * wave + k;
            bf16x8 a[2], b[3];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const char* lo = g1s + off64(y * TW + 4 * g + q, 2 * mt + (p >> 1)) + (p & 1) * 8;
                a[mt] = tr_pair(lo, lo + 16 * 64);
            }
#pragma unroll
            for (int nb = 0; nb < 3; ++nb) {
                const int tap = 4 * nb + p;
                const int ty = tap / 3, tx = tap - 3 * ty;
                const char* lo = tap < 9 ? xs + ((y + ty + 1) * XC + 4 * g + q + tx + 1) * 8 : xs + ZSLOT;
                b[nb] = tr_pair(lo, tap < 9 ? lo + 16 * 8 : lo);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) gacc[mt][nb] = mfma32(a[mt], b[nb], gacc[mt][nb]);
        }
    }
    float* out = partial + ((size_t)blockIdx.x * NW + wave) * PART5;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = row16_sum(sg[nt][e]), b = row16_sum(sgy[nt][e]);
            if (c == 0) {
                out[16 * nt + 4 * g + e] = a;
                out[C1 + 16 * nt + 4 * g + e] = b;
            }
        }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nb = 0; nb < 3; ++nb)
#pragma unroll
            for (int e = 0; e < 4; ++e) out[2 * C1 + (16 * mt + 4 * g + e) * 48 + 16 * nb + c] = gacc[mt][nb][e];
}

inline int grid_for(int ntiles) { return ntiles < 256 ? ntiles : 256; }

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
    long long n = 256ll * NW * PART1;
    const long long a = ((M + TOK_WG - 1) / TOK_WG) * NW * PART3, b = 64ll * 4 * WG3_OUT, c2 = 256ll * WG2_OUT;
    n = n > a ? n : a;
    n = n > b ? n : b;
    n = n > c2 ? n : c2;
    return n > 0x7fffffffll ? PSWIN_ERR_ARG : (int)n;
}

int pswin_stem_conv1_stats(const void* x4, const void* w1p, int B, int H, int W, int want_xx, float* sums,
                           float* workspace, void* stream) {
    PSWIN_CHECK_ARG(x4 && w1p && sums && workspace && B > 0 && H > 0 && W > 0);
    const int nty = (H + TH - 1) / TH, ntx = (W + TW - 1) / TW, ntiles = B * nty * ntx;
    const int grid = grid_for(ntiles);
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
    const int nty = (H + TH - 1) / TH, ntx = (W + TW - 1) / TW, ntiles = B * nty * ntx;
    const int grid = grid_for(ntiles);
    hipLaunchKernelGGL(stem_conv2_fwd_kernel, dim3(grid), dim3(WG), 0, (hipStream_t)stream, reinterpret_cast<const u64*>(x4),
                       w1p, scale1, shift1, w2p, H, W, nty, ntx, ntiles, y2, sums2 ? workspace : nullptr);
    if (sums2) launch_colsum(workspace, grid * NW, PART2, sums2, (hipStream_t)stream);
    PSWIN_LAUNCH_RET();
}

int pswin_stem_conv3_fwd(const void* y2, const float* scale2, const float* shift2, const void* w3p, const float* bias3,
                         int B, int H, int W, void* tokens, void* stream) {
    PSWIN_CHECK_ARG(y2 && scale2 && shift2 && w3p && bias3 && tokens && B > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0);
    const long long M = (long long)B * (H / 4) * (W / 4);
    const unsigned grid = (unsigned)((M + TOK_WG - 1) / TOK_WG);
    hipLaunchKernelGGL(stem_conv3_fwd_kernel, dim3(grid), dim3(WG), 0, (hipStream_t)stream, y2, scale2, shift2, w3p, bias3, H,
                       W, M, (long long)B * H * W * 128, tokens);
    PSWIN_LAUNCH_RET();
}

int pswin_stem_conv3_bwd_stats(const void* dtok, const void* y2, const float* prm, const void* w3t, int B, int H, int W,
                               float* sums, float* workspace, void* stream) {
    PSWIN_CHECK_ARG(dtok && y2 && prm && w3t && sums && workspace && B > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0);
    const long long M = (long long)B * (H / 4) * (W / 4);
    const unsigned grid = (unsigned)((M + TOK_WG - 1) / TOK_WG);
    PSWIN_CHECK_ARG((long long)grid * NW * PART3 <= (long long)pswin_stem_workspace(B, H, W));
    hipLaunchKernelGGL(stem_conv3_bwd_kernel<false>, dim3(grid), dim3(WG), 0, (hipStream_t)stream, dtok, y2, prm, w3t, H, W, M,
                       (long long)B * H * W * 128, nullptr, workspace);
    launch_colsum(workspace, (int)grid * NW, PART3, sums, (hipStream_t)stream);
    PSWIN_LAUNCH_RET();
}

int pswin_stem_conv3_bwd_data(const void* dtok, const void* y2, const float* prm, const void* w3t, int B, int H, int W,
                              void* dy2, void* stream) {
    PSWIN_CHECK_ARG(dtok && y2 && prm && w3t && dy2 && B > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0);
    const long long M = (long long)B * (H / 4) * (W / 4);
    const unsigned grid = (unsigned)((M + TOK_WG - 1) / TOK_WG);
    hipLaunchKernelGGL(stem_conv3_bwd_kernel<true>, dim3(grid), dim3(WG), 0, (hipStream_t)stream, dtok, y2, prm, w3t, H, W, M,
                       (long long)B * H * W * 128, dy2, nullptr);
    PSWIN_LAUNCH_RET();
}

int pswin_stem_conv3_wgrad(const void* dtok, const void* y2, const float* scale2, const float* shift2, int B, int H, int W,
                           float* dw3, float* workspace, void* stream) {
    PSWIN_CHECK_ARG(dtok && y2 && scale2 && shift2 && dw3 && workspace && B > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0);
    const long long M = (long long)B * (H / 4) * (W / 4);
    const long long nsteps = (M + 31) / 32;
    int splits = (int)(nsteps / 4 < 64 ? (nsteps + 3) / 4 : 64);     // workgroups per tap row, >= 4 steps each
    if (splits < 1) splits = 1;
    const int per = (int)((nsteps + splits - 1) / splits);
    PSWIN_CHECK_ARG((long long)splits * 4 * WG3_OUT <= (long long)pswin_stem_workspace(B, H, W));
    hipLaunchKernelGGL(stem_conv3_wgrad_kernel, dim3(splits, 4), dim3(WG), 0, (hipStream_t)stream, dtok, y2, scale2, shift2, H,
                       W, M, per, workspace);
    launch_colsum(workspace, splits, 4 * WG3_OUT, dw3, (hipStream_t)stream);
    PSWIN_LAUNCH_RET();
}

int pswin_stem_conv2_wgrad(const void* x4, const void* w1p, const float* scale1, const float* shift1, const void* dy2, int B,
                           int H, int W, float* dw2, float* workspace, void* stream) {
    PSWIN_CHECK_ARG(x4 && w1p && scale1 && shift1 && dy2 && dw2 && workspace && B > 0 && H > 0 && W > 0);
    const int nty = (H + TH - 1) / TH, ntx = (W + TW - 1) / TW, ntiles = B * nty * ntx;
    const int grid = grid_for(ntiles);
    hipLaunchKernelGGL(stem_conv2_wgrad_kernel, dim3(grid), dim3(WG), 0, (hipStream_t)stream, reinterpret_cast<const u64*>(x4),
                       w1p, scale1, shift1, dy2, H, W, nty, ntx, ntiles, workspace);
    launch_colsum(workspace, grid, WG2_OUT, dw2, (hipStream_t)stream);
    PSWIN_LAUNCH_RET();
}

int pswin_stem_conv2_bwd(const void* x4, const void* w1p, const float* prm, const void* dy2, const void* w2t, int B, int H,
                         int W, float* out, float* workspace, void* stream) {
    PSWIN_CHECK_ARG(x4 && w1p && prm && dy2 && w2t && out && workspace && B > 0 && H > 0 && W > 0);
    const int nty = (H + TH - 1) / TH, ntx = (W + TW - 1) / TW, ntiles = B * nty * ntx;
    const int grid = grid_for(ntiles);
    hipLaunchKernelGGL(stem_conv2_bwd_kernel, dim3(grid), dim3(WG), 0, (hipStream_t)stream, reinterpret_cast<const u64*>(x4),
                       w1p, prm, dy2, w2t, H, W, nty, ntx, ntiles, workspace);
    launch_colsum(workspace, grid * NW, PART5, out, (hipStream_t)stream);
    PSWIN_LAUNCH_RET();
}

}  // extern "C"
