// Per-window fused  qkv Linear -> 7x7 multi-head attention -> proj Linear  for gfx950 (MI355X), C = 96 / 3 heads
// (the high-resolution stage of PanoSwin-T / -S).
//
// Replaces WindowAttention.forward (HOT:274-323: self.qkv(x), q*scale, q@k^T, + great-circle / relative-position bias,
// + mask, softmax, @v, self.proj) for one window per wave.  The unfused path writes the [B*nW*49, 3C] qkv tensor with a
// GEMM, re-reads it in the attention kernel, writes [.., C], re-reads it in the proj GEMM: 8 row passes over HBM for
// an operator whose arithmetic intensity, fused, is ~240 FLOP/B.  Here a window's 49 x 96 input rows are read
// once, its 49 x 96 output rows written once (training mode additionally stores qkv, the attention output and the
// log-sum-exp rows that the backward kernels read), and every product runs on the MFMA pipe out of registers.
// HOT = mmdet/models/backbones/simple_panoswin_transformer.py of the reference.
//
// Design (CDNA4, wave64, one 8-wave workgroup per CU):
//   * the WHOLE weight set lives in LDS for the lifetime of the (persistent) workgroup: Wqkv [288][96] and Wproj [96][96]
//     bf16 as MFMA operand row fragments (72 KB, 16-byte chunks XOR-swizzled inside their group of 4 by row bits 1-2:
//     the 16-byte reads of 16 rows x 4 chunks are conflict-free), + the qkv bias;
//   * a workgroup owns one bias window wb at a time: the score bias of its 3 heads, (d * alpha[idx] + beta[idx] + mask) /
//     scale with -inf in the padded key columns, is built ONCE into LDS as f32 [3][49][64] (quads rotated by 2 * row:
//     conflict-free 16-byte reads) and enters every score tile as the MFMA C operand;
//   * each of the 8 waves then takes the images of the batch that share that window (window n = rep * nb + wb) and runs
//     the whole chain for its window WITHOUT any LDS round trip of activations, by choosing the orientation of every
//     product so that its accumulator is already the next product's operand (a token's 16-byte row chunk IS an operand
//     fragment, and the contraction order inside a 32-deep step is free as long as both operands agree):
//         Q^T, K^T [d][token] = W . X^T     (A = weight rows, B = X rows)   -> packed: B / A operand of the scores
//         V        [token][d] = X . Wv^T    (A = X rows, B = weight rows)   -> packed: A operand (V^T) of P.V
//         S^T      [key][query] = K . Q^T + bias       softmax over the accumulator rows of one lane (+ 2 lane swaps)
//         O^T      [d][query] = V^T . P^T              -> packed: B operand of the projection
//         Y^T      [f][query] = Wproj . O^T            (Wproj staged with its contraction index permuted to match)
//     384 MFMAs (v_mfma_f32_16x16x32_bf16) per window against 18 KB of HBM traffic;
//   * training mode (SAVE): q, k, v rows, the attention output rows and the log-sum-exp go to HBM for pswin_attn_bwd /
//     the weight-gradient GEMMs; V is then produced in the token-on-lane orientation (so that it can be stored as rows)
//     and reaches the P.V product through a 4 KB per-wave LDS image read back with ds_read_b64_tr_b16.
// Rounding points are those of the unfused bf16 path: qkv and the attention output are rounded to bf16 before they are
// used as operands; scores, softmax and all accumulation are f32.
#include "pswin_attn_frag.hpp"

using namespace pswin;

namespace {

constexpr int FC = 96;                 // channels
constexpr int FH = FC / HD;            // 3 heads
constexpr int FKS = FC / 32;           // 32-deep contraction steps of the projections
constexpr int FWAVES = 8;
constexpr int FTHREADS = 64 * FWAVES;

constexpr int WQ_BYTES = 3 * FC * FC * 2;            // 55,296
constexpr int WP_BYTES = FC * FC * 2;                // 18,432
constexpr int BQ_BYTES = 3 * FC * 4;                 //  1,152
constexpr int TAB_BYTES = 2 * FH * TABP * 4;         //  4,224  alpha / beta columns of the 3 heads
constexpr int BIAS_BYTES = FH * TOK * PADT * 4;      // 37,632  [head][query][64 keys] f32, double buffered

struct FusedArgs {
    const void* x;          // [n*49][96] bf16 window rows
    const void* wqkv;       // [288][96] bf16
    const float* bqkv;      // [288] or null
    const void* wproj;      // [96][96] bf16
    const float* dist;      // [n_dist][64][64] f32 tiles (pswin_attn_pad_tiles, not transposed) or null
    const float* mask;      // same layout or null
    const float* alpha;     // [169][3]
    const float* beta;      // [169][3]
    void* y;                // [n*49][96] bf16: attention output @ Wproj^T (no bias)
    void* qkv;              // SAVE: [n][3 heads][q | k | v][49][32] bf16
    void* att;              // SAVE: [n*49][96] bf16
    float* lse;             // SAVE: [n][3][64] f32
    int n_dist, n_mask, nb, reps;
    float scale;
};

__device__ inline int woff(int row, int chunk) { return row * 192 + (((chunk & ~3) | ((chunk ^ (row >> 1)) & 3)) << 4); }
__device__ inline int bias_off(int h, int i, int q) { return ((h * TOK + i) * 16 + ((q + 2 * i) & 15)) * 16; }   // bytes

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
// one v_cvt_pk_bf16_f32 (the scalar cast + shift + or form costs 2 conversions and 2 integer ops per pair)
__device__ inline unsigned pk(float lo, float hi) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2)); }
// two accumulator quads (elements 4g+e of two 16-wide tiles) -> the 8-element operand fragment {lo[0..3], hi[0..3]}
__device__ inline u32x4 pack8(f32x4 lo, f32x4 hi) { return u32x4{pk(lo[0], lo[1]), pk(lo[2], lo[3]), pk(hi[0], hi[1]), pk(hi[2], hi[3])}; }
// the same 8 values of this lane and of the lane 16 away re-grouped into 8 CONTIGUOUS columns starting at
// 8 (g >> 1) + 16 (g & 1) of the 32-column group (one 16-byte row store instead of two 8-byte ones)
__device__ inline u32x4 row8(u32x4 f) {
    const auto r0 = __builtin_amdgcn_permlane16_swap(f[0], f[2], false, false);     // the builtin: hipcc pads the hazard only where needed
    const auto r1 = __builtin_amdgcn_permlane16_swap(f[1], f[3], false, false);
    return u32x4{r0[0], r1[0], r0[1], r1[1]};
}
__device__ inline f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

#ifdef PSWIN_FUSED_CLOCK_PROBE
__device__ unsigned long long pswin_fused_clock_probe[256][2];
#endif

template <bool SAVE>
__global__ __launch_bounds__(FTHREADS, 2) void win_fused_fwd_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wq = smem;
    char* wp = wq + WQ_BYTES;
    float* bq = reinterpret_cast<float*>(wp + WP_BYTES);
    float* tabs = reinterpret_cast<float*>(reinterpret_cast<char*>(bq) + BQ_BYTES);     // [head][alpha | beta][TABP]
    char* bias = reinterpret_cast<char*>(tabs) + TAB_BYTES;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int c = lane & 15, g = lane >> 4;

    // ---- once per workgroup: weights, qkv bias and the table columns -> LDS --------------------------------------
    for (int i = tid; i < 3 * FC * (FC / 8); i += FTHREADS) {
        const int row = i / (FC / 8), ch = i - row * (FC / 8);
        *reinterpret_cast<u32x4*>(wq + woff(row, ch)) = reinterpret_cast<const u32x4*>(a.wqkv)[i];
    }
    // Wproj: contraction index k = 32 h + 16 dt + 4 q + e is stored at chunk 4 h + q, element 4 dt + e, the order in which a
    // lane holds the attention output after packing its two accumulator tiles (dt = 0, 1) of head h
    for (int i = tid; i < FC * (FC / 8); i += FTHREADS) {
        const int row = i / (FC / 8), ch = i - row * (FC / 8);
        const u32x4 v = reinterpret_cast<const u32x4*>(a.wproj)[i];
        const int h = ch >> 2, cc = ch & 3, dt = cc >> 1, q0 = 2 * (cc & 1);
        *reinterpret_cast<u32x2*>(wp + woff(row, 4 * h + q0) + 8 * dt) = u32x2{v[0], v[1]};
        *reinterpret_cast<u32x2*>(wp + woff(row, 4 * h + q0 + 1) + 8 * dt) = u32x2{v[2], v[3]};
    }
    for (int i = tid; i < 3 * FC; i += FTHREADS) bq[i] = a.bqkv ? a.bqkv[i] : 0.f;
    for (int i = tid; i < FH * TABP; i += FTHREADS) {
        const int h = i / TABP, t = i - h * TABP;
        tabs[(2 * h) * TABP + t] = (a.dist && t < NBINS) ? a.alpha[t * FH + h] : 0.f;
        tabs[(2 * h + 1) * TABP + t] = t < NBINS ? a.beta[t * FH + h] : 0.f;
    }
    const float inv_scale = 1.0f / a.scale;
    const float sl2e = a.scale * LOG2E;
    const int d0 = 8 * (g >> 1) + 16 * (g & 1);       // first of this lane's 8 contiguous columns after row8()
    // LDS addresses as ONE per-lane base + compile-time offsets (ds_read offset field), instead of one hoisted VGPR per
    // (row block, step): for row blocks that start at a multiple of 8, woff(R + c, 4 s + g) = R * 192 + 64 s + wlane
    const char* wq_l = wq + c * 192 + (((g ^ (c >> 1)) & 3) << 4);
    const char* wp_l = wp + c * 192 + (((g ^ (c >> 1)) & 3) << 4);
    // bias_off(h, 16 tq + c, 4 tj + g) = (h * 49 + 16 tq) * 256 + blane[tj]  (the quad rotation wraps per lane)
    int blane[4];
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) blane[tj] = c * 256 + (((4 * tj + g + 2 * c) & 15) << 4);

    // score bias of (wb, all heads) / scale -> LDS buffer `dst`, query-major, quads along the keys
    auto build_bias = [&](int wb, char* dst) {
        const float* dtile = a.dist ? a.dist + (size_t)(wb % a.n_dist) * (PADT * PADT) : nullptr;
        const float* mtile = a.mask ? a.mask + (size_t)(wb % a.n_mask) * (PADT * PADT) : nullptr;
        // one (query i, key quad q) per thread and iteration: the index / distance / mask quad is shared by the heads
#pragma unroll 1
        for (int t = tid; t < TOK * 16; t += FTHREADS) {
            const int i = t >> 4, q = t & 15;
            const f32x4 d4 = dtile ? *reinterpret_cast<const f32x4*>(dtile + i * PADT + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 m4 = mtile ? *reinterpret_cast<const f32x4*>(mtile + i * PADT + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            int idx[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) idx[e] = (4 * q + e < TOK) ? rel_a(i) - rel_b(4 * q + e) : 0;
#pragma unroll
            for (int h = 0; h < FH; ++h) {
                const float* ta = tabs + (2 * h) * TABP;
                const float* tb = ta + TABP;
                f32x4 r;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // same rounding sequence as the reference: (d * alpha + beta) [+ mask]   (HOT:255-256, 294, 301)
                    float val = tb[idx[e]];
                    if (dtile) val = __fadd_rn(__fmul_rn(d4[e], ta[idx[e]]), val);
                    if (mtile) val = __fadd_rn(val, m4[e]);
                    r[e] = (4 * q + e < TOK) ? val * inv_scale : -INFINITY;      // padded key: never receives weight
                }
                *reinterpret_cast<f32x4*>(dst + bias_off(h, i, q)) = r;
            }
        }
    };
    // X rows of window (rep, wb) as operand fragments: tile t = tokens 16 t + c, step s = channels 32 s + 8 g ..; rows >= 49
    // read zeros (buffer range check)
    u32x4 xb[4][FKS];
    auto load_x = [&](int rep, int wb) {
        const size_t row0 = ((size_t)rep * a.nb + wb) * TOK;
        const rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.x)) + row0 * (FC * 2), 0,
                                                            TOK * FC * 2, 0x00020000);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int s = 0; s < FKS; ++s)
                xb[t][s] = __builtin_amdgcn_raw_buffer_load_b128(xs, (unsigned)((16 * t + c) * (FC * 2) + 64 * s + 16 * g), 0, 0);
    };

    // The bias is double buffered and built one window ahead, and a wave's first X tile of the next window is requested
    // before that build: ONE barrier per window, with the build and the HBM latency of X on the far side of it.
    __syncthreads();                                  // tables staged
    if (blockIdx.x < a.nb) {
        if (wave < a.reps) load_x(wave, blockIdx.x);
        build_bias(blockIdx.x, bias);
    }
    __syncthreads();
    int par = 0;
#ifdef PSWIN_FUSED_CLOCK_PROBE
    // diagnostic build only (tools/probe/clock_probe.hip): shader-clock ticks and 100 MHz reference ticks around the window loop of
    // every workgroup -> the clock the chip holds under this kernel's load (MI355X_MICROARCH.md, DVFS give-back item 6).  The stamps
    // go to a buffer of their own; no output depends on them.
    const unsigned long long pswin_t0 = __builtin_amdgcn_s_memtime(), pswin_r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    for (int wb = blockIdx.x; wb < a.nb; wb += gridDim.x, par ^= 1) {
        const char* bcur = bias + par * BIAS_BYTES;
        for (int rep = wave; rep < a.reps; rep += FWAVES) {
            if (rep != wave) load_x(rep, wb);         // (batches larger than the 8 waves: loaded in place)
            const size_t win = (size_t)rep * a.nb + wb;
            const size_t row0 = win * TOK;
            const rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.y) + row0 * (FC * 2), 0, TOK * FC * 2, 0x00020000);

            u32x4 of[FH][4];                          // attention output of head h as proj operand fragments [query tile]
            // Weight fragments are read from LDS one projection AHEAD of the MFMAs that consume them (two register sets of
            // 6 x 16 bytes): with 2 waves per SIMD an LDS round trip in front of every 4 MFMAs was half of the wave's life
            // (SQ_WAIT_ANY 49 %).  The scheduling barriers keep hipcc from sinking the reads back to their first use.
            u32x4 wA[2 * FKS], wB[2 * FKS];
            f32x4 bA[2], bB[2];                       // the projection's bias quads (token-on-lane orientation) travel with the set
            auto read_w = [&](int rbase, u32x4 (&w)[2 * FKS], f32x4 (&b)[2]) {
#pragma unroll
                for (int s = 0; s < FKS; ++s)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) w[2 * s + dt] = *reinterpret_cast<const u32x4*>(wq_l + (rbase + 16 * dt) * 192 + 64 * s);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) b[dt] = *reinterpret_cast<const f32x4*>(bq + rbase + 16 * dt + 4 * g);
            };
            // [d = 16 dt + 4 g + e][token 16 t + c] = W rows . X^T + bias  -> operand fragments per token tile
            auto gemm_T = [&](const u32x4 (&w)[2 * FKS], const f32x4 (&b)[2], u32x4 (&frag)[4]) {
                f32x4 acc[2][4];
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[dt][t] = b[dt];
#pragma unroll
                for (int s = 0; s < FKS; ++s)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[dt][t] = mfma(w[2 * s + dt], xb[t][s], acc[dt][t]);
#pragma unroll
                for (int t = 0; t < 4; ++t) frag[t] = pack8(acc[0][t], acc[1][t]);
            };
#pragma unroll
            for (int h = 0; h < FH; ++h) {
                // rows of Wqkv: q = h*32.., k = 96 + h*32.., v = 192 + h*32..
                u32x4 qf[4], kf[4], vt[2][2];
                read_w(h * HD, wA, bA);
                read_w(FC + h * HD, wB, bB);
                const float bv0 = bq[2 * FC + h * HD + c], bv1 = bq[2 * FC + h * HD + 16 + c];   // V bias, feature-on-lane
                __builtin_amdgcn_sched_barrier(0);
                gemm_T(wA, bA, qf);                   // Q^T
                __builtin_amdgcn_sched_barrier(0);
                read_w(2 * FC + h * HD, wA, bA);
                __builtin_amdgcn_sched_barrier(0);
                gemm_T(wB, bB, kf);                   // K^T
                __builtin_amdgcn_sched_barrier(0);
                [[maybe_unused]] rsrc_t vs_save;
                if constexpr (SAVE) {
                    // q, k, v of (window, head) as three contiguous [49][32] blocks ([n][heads][3][49][32], the packing
                    // pswin_attn_bwd_ex reads): a store instruction then writes ONE 1 KB run (16 tokens x 64 B) instead of 16
                    // 64-byte segments 576 B apart; tokens >= 49 fall outside the resource and are dropped
                    constexpr int BLK = TOK * HD * 2;
                    const rsrc_t qs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.qkv) + (win * FH + h) * (size_t)(3 * BLK), 0,
                                                                        BLK, 0x00020000);
                    const rsrc_t ks = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.qkv) + (win * FH + h) * (size_t)(3 * BLK) + BLK, 0,
                                                                        BLK, 0x00020000);
                    vs_save = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.qkv) + (win * FH + h) * (size_t)(3 * BLK) + 2 * BLK, 0, BLK,
                                                                0x00020000);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const unsigned ro = (unsigned)((16 * t + c) * (HD * 2) + d0 * 2);
                        __builtin_amdgcn_raw_buffer_store_b128(row8(qf[t]), qs, ro, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(row8(kf[t]), ks, ro, 0, 0);
                    }
                }
                {
                    // V [token 16 t + 4 g + e][d = 16 dt + c]: X rows as the A operand
                    f32x4 acc[4][2];
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const float b = dt ? bv1 : bv0;
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[t][dt] = f32x4{b, b, b, b};
                    }
#pragma unroll
                    for (int s = 0; s < FKS; ++s)
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                            for (int t = 0; t < 4; ++t) acc[t][dt] = mfma(xb[t][s], wA[2 * s + dt], acc[t][dt]);
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) vt[s][dt] = pack8(acc[2 * s][dt], acc[2 * s + 1][dt]);
                }
                if constexpr (SAVE) {
                    // V rows for the backward pass from the one orientation computed above (round 3; before: a second, transposed V
                    // product, 24 MFMAs per head and the registers that made this variant spill): the packed P.V operand vt[s][dt] IS
                    // V^T as an MFMA A operand, and V^T . E_t with E_t the 0 / 1 matrix that picks token tile t's 16 keys of pair t >> 1
                    // (in pack8's slot order) lands token 16 t + c's features on lane c -- the layout of q and k.  Exact.
                    // key slot 8 g + j of a pair = token 4 g + j (j < 4, even tile) or 16 + 4 g + j - 4 (odd tile); lane (c, g) supplies
                    // column c.  Built here from an opaque copy of the lane index: as loop invariants the two matrices cost 8 registers
                    // across the whole window loop (spills).
                    int cz = c;
                    asm volatile("" : "+v"(cz));
                    u32x4 et_sel[2];
#pragma unroll
                    for (int odd = 0; odd < 2; ++odd)
#pragma unroll
                        for (int j2 = 0; j2 < 4; ++j2) {
                            const int e0 = 2 * (j2 & 1);
                            const bool mine = (j2 >> 1) == odd;
                            et_sel[odd][j2] = ((mine && 4 * g + e0 == cz) ? 0x3F80u : 0u) | ((mine && 4 * g + e0 + 1 == cz) ? 0x3F800000u : 0u);
                        }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        f32x4 vr[2];
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) vr[dt] = mfma(vt[t >> 1][dt], et_sel[t & 1], f32x4{0.f, 0.f, 0.f, 0.f});
                        __builtin_amdgcn_raw_buffer_store_b128(row8(pack8(vr[0], vr[1])), vs_save, (unsigned)((16 * t + c) * (HD * 2) + d0 * 2), 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // (SAVE) attention output rows of this head; without SAVE the resource is never used
                const rsrc_t as = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(SAVE ? a.att : a.y) + row0 * (FC * 2) + h * HD * 2, 0,
                                                                    (TOK - 1) * FC * 2 + HD * 2, 0x00020000);
                // ---- one query tile at a time: S^T, softmax, O^T (16 live score registers instead of 64) -------------------
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) {
                    // S^T [key 16 tj + 4 g + e][query 16 tq + c] = K . Q^T + bias / scale
                    f32x4 s4[4];
                    // (query rows >= 49 read the rows behind their head's 49: in bounds, any value, results discarded)
#pragma unroll
                    for (int tj = 0; tj < 4; ++tj) s4[tj] = *reinterpret_cast<const f32x4*>(bcur + (h * TOK + 16 * tq) * 256 + blane[tj]);
#pragma unroll
                    for (int tj = 0; tj < 4; ++tj) s4[tj] = mfma(kf[tj], qf[tq], s4[tj]);
                    // softmax over the keys of the lane's query (the lane + its 3 partners 16 lanes apart)
                    float mm = s4[3][0];              // key tile 3 holds only key 48 (element 0 of group 0): the rest is -inf
#pragma unroll
                    for (int tj = 0; tj < 3; ++tj)
#pragma unroll
                        for (int e = 0; e < 4; ++e) mm = fmaxf(mm, s4[tj][e]);
                    mm = group_max(mm);
                    const float mb = -mm * sl2e;
#pragma unroll
                    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            s4[tj][e] = (tj == 3 && e > 0) ? 0.f : __builtin_amdgcn_exp2f(__builtin_fmaf(s4[tj][e], sl2e, mb));
                    // O^T [d = 16 dt + 4 g + e][query] = V^T . P^T; a third "d tile" of ones gives the softmax denominator (of
                    // the bf16-rounded weights the product uses) in every accumulator row of the lane's query: no 52-term
                    // VALU sum, no cross-lane step
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
                    of[h][tq] = pack8(o[0] * inv_l, o[1] * inv_l);
                    if constexpr (SAVE) {
                        const int i = 16 * tq + c;
                        __builtin_amdgcn_raw_buffer_store_b128(row8(of[h][tq]), as, (unsigned)(i * (FC * 2) + d0 * 2), 0, 0);
                        if (g == 0)
                            a.lse[(win * FH + h) * PADT + i] = (i < TOK) ? __builtin_fmaf(mm, a.scale, logf(lsum)) : INFINITY;
                    }
                }
            }
            // ---- Y^T [f = 16 ft + 4 g + e][query] = Wproj . O^T, two column tiles at a time -------------------------------
            {
                u32x4 pw[2][2 * FH];                  // Wproj fragments of a pair of column tiles, read one pair ahead
                auto read_p = [&](int np, u32x4 (&w)[2 * FH]) {
#pragma unroll
                    for (int h = 0; h < FH; ++h)
#pragma unroll
                        for (int j = 0; j < 2; ++j) w[2 * h + j] = *reinterpret_cast<const u32x4*>(wp_l + 16 * (2 * np + j) * 192 + 64 * h);
                };
                read_p(0, pw[0]);
#pragma unroll
                for (int np = 0; np < FC / 32; ++np) {
                    if (np + 1 < FC / 32) read_p(np + 1, pw[(np + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 acc[2][4];
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int tq = 0; tq < 4; ++tq) acc[j][tq] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int h = 0; h < FH; ++h)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int tq = 0; tq < 4; ++tq) acc[j][tq] = mfma(pw[np & 1][2 * h + j], of[h][tq], acc[j][tq]);
#pragma unroll
                    for (int tq = 0; tq < 4; ++tq)
                        __builtin_amdgcn_raw_buffer_store_b128(row8(pack8(acc[0][tq], acc[1][tq])), ys,
                                                               (unsigned)((16 * tq + c) * (FC * 2) + 64 * np + d0 * 2), 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        const int wn = wb + gridDim.x;
        if (wn < a.nb) {
            if (wave < a.reps) load_x(wave, wn);
            build_bias(wn, bias + (par ^ 1) * BIAS_BYTES);
        }
        __syncthreads();
    }
#ifdef PSWIN_FUSED_CLOCK_PROBE
    if (threadIdx.x == 0) {
        pswin_fused_clock_probe[blockIdx.x][0] = __builtin_amdgcn_s_memtime() - pswin_t0;
        pswin_fused_clock_probe[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime() - pswin_r0;
    }
#endif
}

// + 15 rows: the padded query rows of the last head read past its 49 rows
constexpr size_t fused_lds(bool) { return WQ_BYTES + WP_BYTES + BQ_BYTES + TAB_BYTES + 2 * BIAS_BYTES + 15 * 256; }

template <bool SAVE>
int launch_fused(const FusedArgs& a, hipStream_t st) {
    static std::atomic<unsigned long long> configured{0};
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&win_fused_fwd_kernel<SAVE>), fused_lds(SAVE), configured)) return rc;
    int grid = a.nb < 256 ? a.nb : 256;              // one persistent workgroup per CU
    hipLaunchKernelGGL((win_fused_fwd_kernel<SAVE>), dim3(grid), dim3(FTHREADS), fused_lds(SAVE), st, a);
    PSWIN_LAUNCH_RET();
}

}  // namespace

extern "C" int pswin_win_attn_fused_supported(int C, int heads, int dtype) { return C == FC && heads == FH && dtype == PSWIN_BF16; }

extern "C" int pswin_win_attn_fused_fwd(const void* x, const void* w_qkv, const float* b_qkv, const void* w_proj, const float* dist_tiles,
                                        int n_dist, const float* alpha, const float* beta, const float* mask_tiles, int n_mask, void* y,
                                        void* qkv_out, void* att_out, float* lse_out, long long n_windows, int n_bias_windows, int C,
                                        int heads, float scale, int dtype, void* stream) {
    PSWIN_CHECK_ARG(x && w_qkv && w_proj && beta && y && n_windows > 0 && n_bias_windows > 0 && scale > 0.f);
    if (!pswin_win_attn_fused_supported(C, heads, dtype)) return PSWIN_ERR_UNSUPPORTED;
    PSWIN_CHECK_ARG(n_windows % n_bias_windows == 0 && n_windows * (long long)(TOK * 3 * FC * 2) < 0x7fffffff00ll);
    PSWIN_CHECK_ARG((dist_tiles == nullptr) == (n_dist == 0) && (mask_tiles == nullptr) == (n_mask == 0));
    PSWIN_CHECK_ARG(!dist_tiles || alpha);
    const bool save = qkv_out || att_out || lse_out;
    PSWIN_CHECK_ARG(!save || (qkv_out && att_out && lse_out));
    PSWIN_CHECK_ARG(aligned16(x) && aligned16(w_qkv) && aligned16(w_proj) && aligned16(y) && aligned16(qkv_out) && aligned16(att_out));
    FusedArgs a;
    a.x = x; a.wqkv = w_qkv; a.bqkv = b_qkv; a.wproj = w_proj; a.dist = dist_tiles; a.mask = mask_tiles; a.alpha = alpha; a.beta = beta;
    a.y = y; a.qkv = qkv_out; a.att = att_out; a.lse = lse_out;
    a.n_dist = n_dist; a.n_mask = n_mask; a.nb = n_bias_windows; a.reps = (int)(n_windows / n_bias_windows); a.scale = scale;
    return save ? launch_fused<true>(a, (hipStream_t)stream) : launch_fused<false>(a, (hipStream_t)stream);
}
