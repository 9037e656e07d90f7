// Per-window fused  qkv Linear -> 7x7 attention core  for the stages behind the first one (gfx950): C = 192 (6 heads) and 384 (12).
//
// Round 3; the sibling of pswin_fused.hip (C = 96, which also owns the proj Linear).  Replaces self.qkv(x) and the attention core of
// WindowAttention.forward (HOT:287-308 of mmdet/models/backbones/simple_panoswin_transformer.py) for one (window, head) per wave:
// the [B*nW*49, 3C] qkv tensor is never written by a GEMM and re-read by an attention kernel -- a window's 49 x C input rows go
// through the MFMA pipe against the head's 96 weight rows, and q, k, v meet the score / softmax / P.V chain of pswin_fused.hip in
// registers.  The proj Linear stays a GEMM: its accumulators (49 x C f32 per window) do not fit beside the rest at C >= 192.
//
// Design (CDNA4, wave64, one 8-wave workgroup per CU):
//   * a workgroup owns ONE head for its lifetime: the head's q, k and v weight rows (96 x C bf16 = 36 / 72 KB) sit in LDS as MFMA
//     operand row fragments -- rows pitched to a multiple of 256 B with the 16-byte chunk XOR-ed by (row & 15): the reads of 16 rows
//     x 4 chunks are conflict-free for ds_read_b128's lane groups -- and it walks the bias windows wb = first, first + stride, ...;
//   * per window the head's score bias (d * alpha[idx] + beta[idx] + mask) / scale is built once into LDS (double buffered, one window
//     ahead, as pswin_fused.hip) and shared by the 8 waves = the images of the batch that share the window;
//   * a wave streams its window's rows from global memory in 32-channel steps (a token's 16-byte row chunk IS an operand fragment)
//     and accumulates Q^T, K^T (A = weight rows) and V (A = X rows) of the head at once: 24 MFMAs per step, the rows three steps
//     ahead in a register ring that runs on into the wave's next window, weight fragments double buffered; what follows -- S^T = K.Q^T + bias as the MFMA C
//     operand, lane-local softmax, O^T = V^T.P^T with the denominator from a "ones" tile -- is pswin_fused.hip's chain, unchanged;
//   * output: the attention rows [n*49][C] (the proj GEMM's input); training mode (SAVE) also stores q, k, v of (window, head) as
//     packed [49][32] blocks and the log-sum-exp rows for pswin_attn_bwd_ex.
// Rounding points are those of the unfused bf16 path: q, k, v and the attention output are rounded to bf16 before they are used as
// operands / stored; scores, softmax and all accumulation are f32.

#include "pswin_attn_frag.hpp"

using namespace pswin;

namespace {

constexpr int QWAVES = 8, QTHREADS = 64 * QWAVES;
using rsrc_t = __amdgpu_buffer_rsrc_t;

struct QkvAttnArgs {
    const void* x;          // [n*49][C] bf16 window rows
    const void* wqkv;       // [3C][C] bf16
    const float* bqkv;      // [3C] or null
    const float* dist;      // [n_dist][64][64] f32 tiles (pswin_attn_pad_tiles, not transposed) or null
    const float* mask;      // same layout or null
    const float* alpha;     // [169][heads]
    const float* beta;      // [169][heads]
    void* y;                // [n*49][C] bf16 attention output
    void* qkv;              // SAVE: [n][heads][q | k | v][49][32] bf16
    float* lse;             // SAVE: [n][heads][64] f32
    int n_dist, n_mask, nb, reps, heads;
    float scale;
    int group;              // bias windows a workgroup has in LDS at a time (round 4): 1, or 8 / reps' for small batches
};

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ inline unsigned pk(float lo, float hi) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2)); }
__device__ inline u32x4 pack8(f32x4 lo, f32x4 hi) { return u32x4{pk(lo[0], lo[1]), pk(lo[2], lo[3]), pk(hi[0], hi[1]), pk(hi[2], hi[3])}; }
__device__ inline u32x4 row8(u32x4 f) {
    const auto r0 = __builtin_amdgcn_permlane16_swap(f[0], f[2], false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(f[1], f[3], false, false);
    return u32x4{r0[0], r1[0], r0[1], r1[1]};
}
__device__ inline f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ inline int bias_off1(int i, int q) { return (i * 16 + ((q + 2 * i) & 15)) * 16; }   // bytes: [query][key quad], quads rotated by 2 * query

#ifdef PSWIN_QA_PROBE
// diagnostic build only (tools/probe/qa_probe.hip): 100 MHz reference ticks at the phase boundaries of every wave, kept in LDS behind
// the kernel's own image and dumped at the end; no output depends on them
constexpr int QA_SLOTS = 40;
__device__ unsigned long long pswin_qa_probe[256][QWAVES][QA_SLOTS];
#define QA_STAMP(k)                                                                             \
    do {                                                                                        \
        const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                         \
        if (lane == 0 && (k) < QA_SLOTS) qa_st[(k)] = t_;                                       \
    } while (0)
#else
#define QA_STAMP(k) do {} while (0)
#endif

template <int C>
struct QGeom {
    static constexpr int KS = C / 32;                              // 32-deep contraction steps
    static constexpr int PITCH = (C * 2 + 255) / 256 * 256;        // weight row pitch: 512 (C = 192) / 768 (C = 384) bytes
    static constexpr int W_BYTES = 96 * PITCH;
    static constexpr int BQ_BYTES = 96 * 4;
    static constexpr int TAB_BYTES = 2 * TABP * 4;
    static constexpr int BIAS_BYTES = TOK * PADT * 4;              // one head: [query][64 keys] f32
    static constexpr int LDS = W_BYTES + BQ_BYTES + TAB_BYTES + 2 * BIAS_BYTES + 15 * 256;      // + 15 rows: padded query rows read past 49
    // Round 4, small batches: with `reps` < 8 images per bias window only `reps` of the 8 waves had an item.  A workgroup now keeps
    // GROUP windows' bias tables in LDS at once (double buffered like the single one) and its waves are (window of the group, image)
    // pairs: batch 2 fills all 8 waves at C = 192 (4 windows x 2 images) and 4 at C = 384 (the 72 KB of weight rows leave room for 2).
    static constexpr int GROUP_MAX = (160 * 1024 - W_BYTES - BQ_BYTES - TAB_BYTES - 15 * 256) / (2 * BIAS_BYTES) >= 4 ? 4 : 2;
    static constexpr int lds_bytes(int group) { return W_BYTES + BQ_BYTES + TAB_BYTES + 2 * group * BIAS_BYTES + 15 * 256; }
    static_assert(KS % 2 == 0, "the channel loop is unrolled by two");
};

template <int C, bool SAVE>
__global__ __launch_bounds__(QTHREADS, 2) void qkv_attn_fwd_kernel(QkvAttnArgs a) {
    using G = QGeom<C>;
    constexpr int KS = G::KS, PITCH = G::PITCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wl = smem;
    float* bq = reinterpret_cast<float*>(wl + G::W_BYTES);         // [q | k | v][32]
    float* tabs = reinterpret_cast<float*>(reinterpret_cast<char*>(bq) + G::BQ_BYTES);      // [alpha | beta][TABP]
    char* bias = reinterpret_cast<char*>(tabs) + G::TAB_BYTES;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
#ifdef PSWIN_QA_PROBE
    unsigned long long* qa_st = reinterpret_cast<unsigned long long*>(smem + G::lds_bytes(a.group)) + wave * QA_SLOTS;
    if (lane < QA_SLOTS) qa_st[lane] = 0;
    QA_STAMP(0);
#endif
    const int c = lane & 15, g = lane >> 4;
    const int heads = a.heads;
    const int hh = (int)(blockIdx.x % (unsigned)heads);            // this workgroup's head
    const int wb0 = (int)(blockIdx.x / (unsigned)heads), wstride = (int)(gridDim.x / (unsigned)heads);

    // ---- once per workgroup: the head's weight rows, bias and table columns -> LDS -------------------------------------
    // local row r = 32 p + d (p = 0 q, 1 k, 2 v) <- Wqkv row p * C + 32 hh + d; chunk ch of a row lives at chunk ch ^ (r & 15).
    // Every global request of the prologue (weight rows, table columns, the first window's distance / mask quads, the first steps of
    // the wave's first item) is issued before the first of them is consumed: one memory round trip instead of four in a row
    // (probe: 6.5 us of a 45 us launch sat in front of the first MFMA).
    constexpr int WCH = 96 * (C / 8), WIT = (WCH + QTHREADS - 1) / QTHREADS;
    u32x4 wst[WIT];
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
        const int i = tid + it * QTHREADS;
        const int r = i / (C / 8), ch = i - r * (C / 8);
        const int src_row = (r >> 5) * C + hh * HD + (r & 31);
        wst[it] = i < WCH ? reinterpret_cast<const u32x4*>(a.wqkv)[(size_t)src_row * (C / 8) + ch] : u32x4{0u, 0u, 0u, 0u};
    }
    float bq_r = 0.f, ta_r = 0.f, tb_r = 0.f;
    if (tid < 96 && a.bqkv) bq_r = a.bqkv[(tid >> 5) * C + hh * HD + (tid & 31)];
    if (tid < NBINS) {
        if (a.dist) ta_r = a.alpha[tid * heads + hh];
        tb_r = a.beta[tid * heads + hh];
    }
    const float inv_scale = 1.0f / a.scale;
    const float sl2e = a.scale * LOG2E;
    const int d0 = 8 * (g >> 1) + 16 * (g & 1);       // first of this lane's 8 contiguous columns after row8()
    // weight fragment (rows R + c, channels 32 s + 8 g ..): R * PITCH + ((4 s + g) ^ c) * 16 with R a multiple of 16.  The XOR only
    // touches the low four chunk bits, so (4 s + g) ^ c = 16 (s >> 2) + ((4 (s & 3) + g) ^ c): four lane bases, the rest immediates
    // (a base per step, as the expression reads, is twelve loop-invariant registers the unrolled loop then spills).
    const char* w_l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w_l[j] = wl + c * PITCH + (((4 * j + g) ^ c) << 4);
    int blane[4];
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) blane[tj] = c * 256 + (((4 * tj + g + 2 * c) & 15) << 4);

    // The score bias of a window is built in two halves, request and finish, so that the prologue can put the first window's
    // distance / mask quads in flight together with the weight rows.  784 quads over 512 threads: two per thread.
    constexpr int BQ_IT = (TOK * 16 + QTHREADS - 1) / QTHREADS;
    f32x4 bd4[BQ_IT], bm4[BQ_IT];
    auto fetch_bias = [&](int wb) {
        const rsrc_t dres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dist ? a.dist + (size_t)(wb % (a.dist ? a.n_dist : 1)) * (PADT * PADT) : nullptr), 0,
                                                              a.dist ? PADT * PADT * 4 : 0, 0x00020000);
        const rsrc_t mres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.mask ? a.mask + (size_t)(wb % (a.mask ? a.n_mask : 1)) * (PADT * PADT) : nullptr), 0,
                                                              a.mask ? PADT * PADT * 4 : 0, 0x00020000);
#pragma unroll
        for (int it = 0; it < BQ_IT; ++it) {
            const int t = tid + it * QTHREADS;
            const unsigned off = t < TOK * 16 ? (unsigned)(((t >> 4) * PADT + 4 * (t & 15)) * 4) : 0xFFFFFF00u;      // no tile / past the tile: zeros
            bd4[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dres, off, 0, 0));
            bm4[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(mres, off, 0, 0));
        }
    };
    auto finish_bias = [&](char* dst) {
        const float* ta = tabs;
        const float* tb = tabs + TABP;
        const bool has_d = a.dist != nullptr, has_m = a.mask != nullptr;
        // the table indices depend only on the thread: left to itself the compiler computes all of them once, in front of the window
        // loop, and carries sixteen registers through the k loop for it.  An opaque zero keeps the arithmetic here.
        int zero = 0;
        asm volatile("" : "+v"(zero));
#pragma unroll
        for (int it = 0; it < BQ_IT; ++it) {
            const int t = tid + it * QTHREADS + zero;
            if (t < TOK * 16) {
                const int i = t >> 4, q = t & 15;
                f32x4 r;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int idx = (4 * q + e < TOK) ? rel_a(i) - rel_b(4 * q + e) : 0;
                    // same rounding sequence as the reference: (d * alpha + beta) [+ mask]   (HOT:255-256, 294, 301)
                    float val = tb[idx];
                    if (has_d) val = __fadd_rn(__fmul_rn(bd4[it][e], ta[idx]), val);
                    if (has_m) val = __fadd_rn(val, bm4[it][e]);
                    r[e] = (4 * q + e < TOK) ? val * inv_scale : -INFINITY;      // padded key: never receives weight
                }
                *reinterpret_cast<f32x4*>(dst + bias_off1(i, q)) = r;
            }
        }
    };

    // ---- the activation ring -------------------------------------------------------------------------------------------
    // A window's rows reach the MFMA pipe straight from global memory, 32 channels (one step = 4 x 16 bytes per lane) at a time.
    // With one step in flight a wave had 4 KB outstanding, the CU 32 KB: at ~2 us of loaded HBM latency that is 16 GB/s per CU, and
    // the kernel ran at exactly that (14.7 GB/s per CU, every step waiting out a full memory round trip).  The ring keeps two steps
    // in flight and runs on across items: the first steps of a wave's NEXT window are requested during the last steps of the
    // current one, so the score / softmax / P.V chain of an item hides the round trip of the next.  Slot of step s of an item that
    // starts at ring phase PH: (PH + s) % XD; the next item starts at (PH + KS) % XD = PH for both geometries.
#ifndef PSWIN_QA_XD_INFER
#define PSWIN_QA_XD_INFER 4
#endif
    // slots: four (one pair in flight behind the pair in use) at C = 384; C = 192 (six steps) takes six, or three single steps in the
    // training variant, which has no registers for six.  Same-box A/B of four against six slots at C = 384: equal within 1 %
    // (the k loop is not waiting on latency any more; MFMA issue, the L1 / address path and LDS reads each sit at 50-100 %).
    constexpr int XD = SAVE ? ((KS % 4 == 0) ? 4 : 3) : (KS % PSWIN_QA_XD_INFER == 0) ? PSWIN_QA_XD_INFER : 6;
    constexpr bool PAIRS = XD % 2 == 0;
    u32x4 xr[6][4];
    const char* xbase = reinterpret_cast<const char*>(a.x);
    auto make_xs = [&](int wb, int rep, bool valid) {
        const size_t row0 = ((size_t)rep * a.nb + wb) * TOK;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xbase) + (valid ? row0 * (C * 2) : 0), 0, valid ? TOK * C * 2 : 0, 0x00020000);
    };
    // X rows as operand fragments: tile t = tokens 16 t + c, step s = channels 32 s + 8 g ..; rows >= 49 (and a null item) read zeros
    const unsigned xlane = (unsigned)(c * (C * 2) + 16 * g);
    auto load_x = [&](rsrc_t xs, int s, u32x4 (&xf)[4]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) xf[t] = __builtin_amdgcn_raw_buffer_load_b128(xs, xlane, 16 * t * (C * 2) + 64 * s, 0);
    };
    // the step's six weight fragments: [q dt0, q dt1, k dt0, k dt1, v dt0, v dt1]
    auto read_w = [&](int s, u32x4 (&w)[6]) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) w[2 * p + dt] = *reinterpret_cast<const u32x4*>(w_l[s & 3] + (32 * p + 16 * dt) * PITCH + (s >> 2) * 256);
    };

    // one (window, head) item of this wave; xs: its rows, xn: the rows of the wave's next item (zero-sized if there is none)
    auto item = [&]<int PH>(size_t win, rsrc_t xs, rsrc_t xn, const char* bcur, [[maybe_unused]] int qa_k) {
        const size_t row0 = win * TOK;
        QA_STAMP(qa_k);
        f32x4 aq[2][4], ak[2][4], av[4][2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const f32x4 bqv = *reinterpret_cast<const f32x4*>(bq + 16 * dt + 4 * g), bkv = *reinterpret_cast<const f32x4*>(bq + 32 + 16 * dt + 4 * g);
            const float bvl = bq[64 + 16 * dt + c];                      // V bias, feature-on-lane
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                aq[dt][t] = bqv;
                ak[dt][t] = bkv;
                av[t][dt] = f32x4{bvl, bvl, bvl, bvl};
            }
        }
        {
            // One set of weight fragments: the MFMAs of a step run fragment by fragment (4 token tiles each), and as soon as the four
            // that read a fragment are issued the same fragment of the NEXT step is requested from LDS into the same registers --
            // 20 MFMAs (320 cycles) ahead of its first use.
            u32x4 w[6];
            read_w(0, w);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                // XD even: steps are requested two at a time, the two 64-byte halves of a row's 128-byte line by consecutive instructions.
                // Requested a step (~0.5 us, 512 other lines per CU) apart, every line was fetched from L2 twice; in pairs the k loop
                // takes the same time with 8 instead of 12 KB in flight per wave and everything else queues less (probe: launch
                // 47.0 -> 42.3 us at C = 384).
                if constexpr (PAIRS) {
                    if ((s & 1) == 0) {
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int h2 = 0; h2 < 2; ++h2) {
                                const int sn = s + XD - 2 + h2;
                                xr[(PH + sn) % XD][t] = __builtin_amdgcn_raw_buffer_load_b128(sn < KS ? xs : xn, xlane, 16 * t * (C * 2) + 64 * (sn < KS ? sn : sn - KS), 0);
                            }
                    }
                } else {
                    const int sn = s + XD - 1;                           // the step requested now
                    if (sn < KS) load_x(xs, sn, xr[(PH + sn) % XD]);
                    else load_x(xn, sn - KS, xr[(PH + sn) % XD]);
                }
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 (&xf)[4] = xr[(PH + s) % XD];
#pragma unroll
                for (int f = 0; f < 6; ++f) {
                    const int p_ = f >> 1, dt = f & 1;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (p_ == 0) aq[dt][t] = mfma(w[f], xf[t], aq[dt][t]);            // Q^T [d][token]
                        else if (p_ == 1) ak[dt][t] = mfma(w[f], xf[t], ak[dt][t]);       // K^T
                        else av[t][dt] = mfma(xf[t], w[f], av[t][dt]);                    // V [token][d]
                    }
                    if (s + 1 < KS)
                        w[f] = *reinterpret_cast<const u32x4*>(w_l[(s + 1) & 3] + (32 * p_ + 16 * dt) * PITCH + ((s + 1) >> 2) * 256);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        QA_STAMP(qa_k + 1);
        u32x4 qf[4], kf[4], vt[2][2];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            qf[t] = pack8(aq[0][t], aq[1][t]);
            kf[t] = pack8(ak[0][t], ak[1][t]);
        }
        // P.V operand: lane (c, g) of vt[s][dt] holds V[keys 32 s + {4 g + e, 16 + 4 g + e}][d = 16 dt + c] -- the key order in which
        // pack8 lays out the probabilities below, so the contraction pairs every key with itself
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) vt[s][dt] = pack8(av[2 * s][dt], av[2 * s + 1][dt]);
        if constexpr (SAVE) {
            constexpr int BLK = TOK * HD * 2;
            char* qb = reinterpret_cast<char*>(a.qkv) + (win * heads + hh) * (size_t)(3 * BLK);
            const rsrc_t qs = __builtin_amdgcn_make_buffer_rsrc(qb, 0, BLK, 0x00020000);
            const rsrc_t ks = __builtin_amdgcn_make_buffer_rsrc(qb + BLK, 0, BLK, 0x00020000);
            const rsrc_t vs = __builtin_amdgcn_make_buffer_rsrc(qb + 2 * BLK, 0, BLK, 0x00020000);
            const unsigned ro = (unsigned)(c * (HD * 2) + d0 * 2);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                __builtin_amdgcn_raw_buffer_store_b128(row8(qf[t]), qs, ro + 16 * t * (HD * 2), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(row8(kf[t]), ks, ro + 16 * t * (HD * 2), 0, 0);
            }
            // V rows for the backward pass from the one V orientation the step computes (a second, transposed set of accumulators
            // cost 8 of 32 MFMAs per step and 32 registers): the bf16 operand vt[s][dt] IS V^T as an MFMA A operand (rows = features,
            // contraction = the 32 keys of pair s in pack8's order), so V^T . E_t with E_t the 0/1 matrix that picks token tile t's 16
            // keys lands token 16 t + c's features on lane c -- the layout of q and k above.  Exact: one product by 1.0 per element.
            // key slot 8 g + j of pair s = token 32 s + (j < 4 ? 4 g + j : 16 + 4 g + j - 4): the even tile of a pair takes the slots
            // j < 4 (operand words 0, 1), the odd tile j >= 4 (words 2, 3); lane (c, g) supplies column c
            u32x4 et[2];
#pragma unroll
            for (int odd = 0; odd < 2; ++odd)
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) {
                    const int e0 = 2 * (j2 & 1);
                    const bool mine = (j2 >> 1) == odd;
                    et[odd][j2] = ((mine && 4 * g + e0 == c) ? 0x3F80u : 0u) | ((mine && 4 * g + e0 + 1 == c) ? 0x3F800000u : 0u);
                }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 vr[2];
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) vr[dt] = mfma(vt[t >> 1][dt], et[t & 1], f32x4{0.f, 0.f, 0.f, 0.f});
                __builtin_amdgcn_raw_buffer_store_b128(row8(pack8(vr[0], vr[1])), vs, ro + 16 * t * (HD * 2), 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        QA_STAMP(qa_k + 2);
        // attention output rows of this head: [49][C] rows, columns 32 hh ..
        const rsrc_t as = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.y) + row0 * (C * 2) + hh * HD * 2, 0, (TOK - 1) * C * 2 + HD * 2,
                                                            0x00020000);
        // ---- one query tile at a time: S^T, softmax, O^T (pswin_fused.hip's chain) -------------------------------------
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            f32x4 s4[4];
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) s4[tj] = *reinterpret_cast<const f32x4*>(bcur + 16 * tq * 256 + blane[tj]);
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) s4[tj] = mfma(kf[tj], qf[tq], s4[tj]);
            float mm = s4[3][0];                  // key tile 3 holds only key 48 (element 0 of group 0): the rest is -inf
#pragma unroll
            for (int tj = 0; tj < 3; ++tj)
#pragma unroll
                for (int e = 0; e < 4; ++e) mm = fmaxf(mm, s4[tj][e]);
            mm = group_max(mm);
            const float mb = -mm * sl2e;
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int e = 0; e < 4; ++e) s4[tj][e] = (tj == 3 && e > 0) ? 0.f : __builtin_amdgcn_exp2f(__builtin_fmaf(s4[tj][e], sl2e, mb));
            f32x4 o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, osum = {0.f, 0.f, 0.f, 0.f};
            const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const u32x4 pf = pack8(s4[2 * s], s4[2 * s + 1]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) o[dt] = mfma(vt[s][dt], pf, o[dt]);
                osum = mfma(ones, pf, osum);
            }
            const float lsum = osum[0];
            const float inv_l = 1.0f / lsum;
            const int i = 16 * tq + c;
            __builtin_amdgcn_raw_buffer_store_b128(row8(pack8(o[0] * inv_l, o[1] * inv_l)), as, (unsigned)(c * (C * 2) + d0 * 2), 16 * tq * (C * 2), 0);
            if constexpr (SAVE) {
                if (g == 0) a.lse[(win * heads + hh) * PADT + i] = (i < TOK) ? __builtin_fmaf(mm, a.scale, logf(lsum)) : INFINITY;
            }
        }
        QA_STAMP(qa_k + 3);
    };

    // prologue, continued: the first window's quads and the first item's first steps join the requests above, then the LDS image
    // Window walk: the workgroup's windows are wb0, wb0 + wstride, ...; it takes GRP of them per round.  GRP == 1 (batches >= 8): the
    // waves are the images of the one window (rep = wave, wave + 8, ...).  GRP > 1: wave = (window gw of the round, image gr).
    const int GRP = a.group, RPW = QWAVES / GRP;                     // images per window a round can take
    const int gw = wave / RPW, gr = wave - gw * RPW;
    if (wb0 < a.nb) fetch_bias(wb0);
    const int first_w = wb0 + gw * wstride;
    rsrc_t xs = make_xs(first_w, gr, first_w < a.nb && gr < a.reps);
#pragma unroll
    for (int s = 0; s < (PAIRS ? XD - 2 : XD - 1); ++s) load_x(xs, s, xr[s]);
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
        const int i = tid + it * QTHREADS;
        const int r = i / (C / 8), ch = i - r * (C / 8);
        if (i < WCH) *reinterpret_cast<u32x4*>(wl + r * PITCH + ((ch ^ (r & 15)) << 4)) = wst[it];
    }
    if (tid < 96) bq[tid] = bq_r;
    if (tid < TABP) {
        tabs[tid] = ta_r;
        tabs[TABP + tid] = tb_r;
    }
    QA_STAMP(1);
    __syncthreads();                                  // weights and tables staged
    QA_STAMP(2);
    if (wb0 < a.nb) finish_bias(bias);
    for (int g = 1; g < GRP; ++g) {                   // the other windows of the first round (small batches only)
        const int wg_ = wb0 + g * wstride;
        if (wg_ < a.nb) {
            fetch_bias(wg_);
            finish_bias(bias + g * G::BIAS_BYTES);
        }
    }
    __syncthreads();
    QA_STAMP(3);
    int par = 0;
    [[maybe_unused]] int qa_it = 0;
    const int rstride = GRP * wstride;                // windows per round
    for (int wb = wb0; wb < a.nb; wb += rstride, par ^= 1) {
        const char* bround = bias + par * GRP * G::BIAS_BYTES;
        const int wn = wb + rstride;
        // ONE call site for both modes (two inlined copies of the item spilled): GRP == 1 gives gw = 0, gr = wave, RPW = 8 -- the image
        // loop of rounds 2-3; GRP > 1 gives at most one trip
        const int wmine = wb + gw * wstride;
        const char* bcur = bround + gw * G::BIAS_BYTES;
        for (int rep = gr; rep < a.reps && wmine < a.nb; rep += RPW) {
            int nrep = rep + RPW, nwb = wmine;
            if (nrep >= a.reps) {
                nrep = gr;
                nwb = wmine + rstride;
            }
            const rsrc_t xn = make_xs(nwb, nrep, nwb < a.nb);
            const size_t win = (size_t)rep * a.nb + wmine;
            static_assert(KS % XD == 0, "every item starts at ring slot 0");
            item.template operator()<0>(win, xs, xn, bcur, 4 + 6 * qa_it);
            xs = xn;
        }
        // (requesting the next window's quads before the score chain and finishing them here was tried: the sixteen registers they
        // hold across the chain spilled the finish's table indices, whose scratch reloads then waited out the item's stores --
        // 1.8 -> 4.5 us for this phase in the probe)
        for (int g = 0; g < GRP; ++g) {
            const int wg_ = wn + g * wstride;
            if (wg_ < a.nb) {
                fetch_bias(wg_);
                finish_bias(bias + ((par ^ 1) * GRP + g) * G::BIAS_BYTES);
            }
        }
        QA_STAMP(4 + 6 * qa_it + 4);
        __syncthreads();
        QA_STAMP(4 + 6 * qa_it + 5);
#ifdef PSWIN_QA_PROBE
        ++qa_it;
#endif
    }
#ifdef PSWIN_QA_PROBE
    if (lane < QA_SLOTS) pswin_qa_probe[blockIdx.x][wave][lane] = qa_st[lane];
#endif
}

template <int C, bool SAVE>
int launch_qkv_attn(QkvAttnArgs a, hipStream_t st) {
    using G = QGeom<C>;
    static std::atomic<unsigned long long> configured{0};
    // windows per round: as many as it takes to give every wave an image (8 / reps, reps rounded up to a power of two), capped by LDS
    int group = 1;
    if (a.reps < QWAVES) {
        int rp = 1;
        while (rp < a.reps) rp *= 2;
        group = QWAVES / rp;
        if (group > G::GROUP_MAX) group = G::GROUP_MAX;
    }
    a.group = group;
#ifdef PSWIN_QA_PROBE
    constexpr int probe_bytes = QWAVES * QA_SLOTS * 8;
#else
    constexpr int probe_bytes = 0;
#endif
    static_assert(G::lds_bytes(G::GROUP_MAX) + probe_bytes <= 160 * 1024, "LDS");
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&qkv_attn_fwd_kernel<C, SAVE>), G::lds_bytes(G::GROUP_MAX) + probe_bytes, configured)) return rc;
    const int lds_bytes = G::lds_bytes(group) + probe_bytes;
    long long items = ((long long)a.nb + group - 1) / group * a.heads;           // (round of windows, head) pairs
    int grid = items < 256 ? (int)items : 256 / a.heads * a.heads;        // a multiple of `heads`: workgroup b owns head b % heads
    hipLaunchKernelGGL((qkv_attn_fwd_kernel<C, SAVE>), dim3(grid), dim3(QTHREADS), lds_bytes, st, a);
    PSWIN_LAUNCH_RET();
}

}  // namespace

extern "C" int pswin_qkv_attn_fused_supported(int C, int heads, int dtype) {
    return (C == 192 || C == 384) && heads * HD == C && dtype == PSWIN_BF16;
}

extern "C" int pswin_qkv_attn_fused_fwd(const void* x, const void* w_qkv, const float* b_qkv, const float* dist_tiles, int n_dist, const float* alpha,
                                        const float* beta, const float* mask_tiles, int n_mask, void* y, void* qkv_out, float* lse_out,
                                        long long n_windows, int n_bias_windows, int C, int heads, float scale, int dtype, void* stream) {
    PSWIN_CHECK_ARG(x && w_qkv && beta && y && n_windows > 0 && n_bias_windows > 0 && scale > 0.f);
    if (!pswin_qkv_attn_fused_supported(C, heads, dtype)) return PSWIN_ERR_UNSUPPORTED;
    PSWIN_CHECK_ARG(n_windows % n_bias_windows == 0 && n_windows * (long long)(TOK * 3 * C * 2) < 0x7fffffff00ll);
    PSWIN_CHECK_ARG((dist_tiles == nullptr) == (n_dist == 0) && (mask_tiles == nullptr) == (n_mask == 0));
    PSWIN_CHECK_ARG(!dist_tiles || alpha);
    const bool save = qkv_out || lse_out;
    PSWIN_CHECK_ARG(!save || (qkv_out && lse_out));
    PSWIN_CHECK_ARG(aligned16(x) && aligned16(w_qkv) && aligned16(y) && aligned16(qkv_out));
    QkvAttnArgs a;
    a.x = x; a.wqkv = w_qkv; a.bqkv = b_qkv; a.dist = dist_tiles; a.mask = mask_tiles; a.alpha = alpha; a.beta = beta;
    a.y = y; a.qkv = qkv_out; a.lse = lse_out;
    a.n_dist = n_dist; a.n_mask = n_mask; a.nb = n_bias_windows; a.reps = (int)(n_windows / n_bias_windows); a.heads = heads; a.scale = scale;
    const hipStream_t st = (hipStream_t)stream;
    if (C == 192) return save ? launch_qkv_attn<192, true>(a, st) : launch_qkv_attn<192, false>(a, st);
    return save ? launch_qkv_attn<384, true>(a, st) : launch_qkv_attn<384, false>(a, st);
}
