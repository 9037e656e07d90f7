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
__device__ inline void swap16_u32(unsigned& a, unsigned& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }

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
// the same exchange on f32 quads (8 consecutive channels as two quads)
__device__ inline void exchange_row8(f32x4& q0, f32x4& q1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned a = __builtin_bit_cast(unsigned, q0[e]), b = __builtin_bit_cast(unsigned, q1[e]);
        swap16_u32(a, b);
        q0[e] = __builtin_bit_cast(float, a);
        q1[e] = __builtin_bit_cast(float, b);
    }
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
    return 256 * NW * PART1;     // the largest per-wave partial set (statistics pass); see also the backward kernels
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

}  // extern "C"
