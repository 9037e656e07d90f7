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
//     and accumulates Q^T, K^T (A = weight rows) and V (A = X rows) of the head at once: 24 MFMAs per step (32 with the training
//     variant's second V orientation), operands double buffered in registers; what follows -- S^T = K.Q^T + bias as the MFMA C
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

template <int C>
struct QGeom {
    static constexpr int KS = C / 32;                              // 32-deep contraction steps
    static constexpr int PITCH = (C * 2 + 255) / 256 * 256;        // weight row pitch: 512 (C = 192) / 768 (C = 384) bytes
    static constexpr int W_BYTES = 96 * PITCH;
    static constexpr int BQ_BYTES = 96 * 4;
    static constexpr int TAB_BYTES = 2 * TABP * 4;
    static constexpr int BIAS_BYTES = TOK * PADT * 4;              // one head: [query][64 keys] f32
    static constexpr int LDS = W_BYTES + BQ_BYTES + TAB_BYTES + 2 * BIAS_BYTES + 15 * 256;      // + 15 rows: padded query rows read past 49
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
    const int c = lane & 15, g = lane >> 4;
    const int heads = a.heads;
    const int hh = (int)(blockIdx.x % (unsigned)heads);            // this workgroup's head
    const int wb0 = (int)(blockIdx.x / (unsigned)heads), wstride = (int)(gridDim.x / (unsigned)heads);

    // ---- once per workgroup: the head's weight rows, bias and table columns -> LDS -------------------------------------
    // local row r = 32 p + d (p = 0 q, 1 k, 2 v) <- Wqkv row p * C + 32 hh + d; chunk ch of a row lives at chunk ch ^ (r & 15)
    for (int i = tid; i < 96 * (C / 8); i += QTHREADS) {
        const int r = i / (C / 8), ch = i - r * (C / 8);
        const int src_row = (r >> 5) * C + hh * HD + (r & 31);
        *reinterpret_cast<u32x4*>(wl + r * PITCH + ((ch ^ (r & 15)) << 4)) = reinterpret_cast<const u32x4*>(a.wqkv)[(size_t)src_row * (C / 8) + ch];
    }
    for (int i = tid; i < 96; i += QTHREADS) bq[i] = a.bqkv ? a.bqkv[(i >> 5) * C + hh * HD + (i & 31)] : 0.f;
    for (int t = tid; t < TABP; t += QTHREADS) {
        tabs[t] = (a.dist && t < NBINS) ? a.alpha[t * heads + hh] : 0.f;
        tabs[TABP + t] = t < NBINS ? a.beta[t * heads + hh] : 0.f;
    }
    const float inv_scale = 1.0f / a.scale;
    const float sl2e = a.scale * LOG2E;
    const int d0 = 8 * (g >> 1) + 16 * (g & 1);       // first of this lane's 8 contiguous columns after row8()
    // weight fragment (rows R + c, channels 32 s + 8 g ..): R * PITCH + ((4 s + g) ^ c) * 16 with R a multiple of 16
    const char* w_l = wl + c * PITCH;
    int blane[4];
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) blane[tj] = c * 256 + (((4 * tj + g + 2 * c) & 15) << 4);

    auto build_bias = [&](int wb, char* dst) {
        const float* dtile = a.dist ? a.dist + (size_t)(wb % a.n_dist) * (PADT * PADT) : nullptr;
        const float* mtile = a.mask ? a.mask + (size_t)(wb % a.n_mask) * (PADT * PADT) : nullptr;
        const float* ta = tabs;
        const float* tb = tabs + TABP;
#pragma unroll 1
        for (int t = tid; t < TOK * 16; t += QTHREADS) {
            const int i = t >> 4, q = t & 15;
            const f32x4 d4 = dtile ? *reinterpret_cast<const f32x4*>(dtile + i * PADT + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 m4 = mtile ? *reinterpret_cast<const f32x4*>(mtile + i * PADT + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = (4 * q + e < TOK) ? rel_a(i) - rel_b(4 * q + e) : 0;
                // same rounding sequence as the reference: (d * alpha + beta) [+ mask]   (HOT:255-256, 294, 301)
                float val = tb[idx];
                if (dtile) val = __fadd_rn(__fmul_rn(d4[e], ta[idx]), val);
                if (mtile) val = __fadd_rn(val, m4[e]);
                r[e] = (4 * q + e < TOK) ? val * inv_scale : -INFINITY;      // padded key: never receives weight
            }
            *reinterpret_cast<f32x4*>(dst + bias_off1(i, q)) = r;
        }
    };

    __syncthreads();                                  // tables staged
    if (wb0 < a.nb) build_bias(wb0, bias);
    __syncthreads();
    int par = 0;
    for (int wb = wb0; wb < a.nb; wb += wstride, par ^= 1) {
        const char* bcur = bias + par * G::BIAS_BYTES;
        for (int rep = wave; rep < a.reps; rep += QWAVES) {
            const size_t win = (size_t)rep * a.nb + wb;
            const size_t row0 = win * TOK;
            const rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.x)) + row0 * (C * 2), 0, TOK * C * 2,
                                                                0x00020000);
            // X rows as operand fragments: tile t = tokens 16 t + c, step s = channels 32 s + 8 g ..; rows >= 49 read zeros
            auto load_x = [&](int s, u32x4 (&xf)[4]) {
#pragma unroll
                for (int t = 0; t < 4; ++t) xf[t] = __builtin_amdgcn_raw_buffer_load_b128(xs, (unsigned)((16 * t + c) * (C * 2) + 64 * s + 16 * g), 0, 0);
            };
            // the step's six weight fragments: [q dt0, q dt1, k dt0, k dt1, v dt0, v dt1]
            auto read_w = [&](int s, u32x4 (&w)[6]) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) w[2 * p + dt] = *reinterpret_cast<const u32x4*>(w_l + (32 * p + 16 * dt) * PITCH + (((4 * s + g) ^ c) << 4));
            };
            f32x4 aq[2][4], ak[2][4], av[4][2];
            [[maybe_unused]] f32x4 avt[2][4];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const f32x4 bqv = *reinterpret_cast<const f32x4*>(bq + 16 * dt + 4 * g), bkv = *reinterpret_cast<const f32x4*>(bq + 32 + 16 * dt + 4 * g);
                const f32x4 bvv = *reinterpret_cast<const f32x4*>(bq + 64 + 16 * dt + 4 * g);
                const float bvl = bq[64 + 16 * dt + c];                      // V bias, feature-on-lane
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    aq[dt][t] = bqv;
                    ak[dt][t] = bkv;
                    av[t][dt] = f32x4{bvl, bvl, bvl, bvl};
                    if constexpr (SAVE) avt[dt][t] = bvv;
                }
            }
            auto mma_step = [&](const u32x4 (&xf)[4], const u32x4 (&w)[6]) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        aq[dt][t] = mfma(w[dt], xf[t], aq[dt][t]);           // Q^T [d][token]
                        ak[dt][t] = mfma(w[2 + dt], xf[t], ak[dt][t]);       // K^T
                        av[t][dt] = mfma(xf[t], w[4 + dt], av[t][dt]);       // V [token][d]
                        if constexpr (SAVE) avt[dt][t] = mfma(w[4 + dt], xf[t], avt[dt][t]);      // V^T: rows for the backward pass
                    }
            };
            {
                u32x4 xa[4], xb[4], wa[6], wbf[6];
                load_x(0, xa);
                read_w(0, wa);
#pragma unroll 1
                for (int s = 0; s < KS; s += 2) {
                    load_x(s + 1, xb);
                    read_w(s + 1, wbf);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_step(xa, wa);
                    __builtin_amdgcn_sched_barrier(0);
                    if (s + 2 < KS) {
                        load_x(s + 2, xa);
                        read_w(s + 2, wa);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    mma_step(xb, wbf);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            u32x4 qf[4], kf[4], vt[2][2];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                qf[t] = pack8(aq[0][t], aq[1][t]);
                kf[t] = pack8(ak[0][t], ak[1][t]);
            }
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
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const unsigned ro = (unsigned)((16 * t + c) * (HD * 2) + d0 * 2);
                    __builtin_amdgcn_raw_buffer_store_b128(row8(qf[t]), qs, ro, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(row8(kf[t]), ks, ro, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(row8(pack8(avt[0][t], avt[1][t])), vs, ro, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
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
                __builtin_amdgcn_raw_buffer_store_b128(row8(pack8(o[0] * inv_l, o[1] * inv_l)), as, (unsigned)(i * (C * 2) + d0 * 2), 0, 0);
                if constexpr (SAVE) {
                    if (g == 0) a.lse[(win * heads + hh) * PADT + i] = (i < TOK) ? __builtin_fmaf(mm, a.scale, logf(lsum)) : INFINITY;
                }
            }
        }
        const int wn = wb + wstride;
        if (wn < a.nb) build_bias(wn, bias + (par ^ 1) * G::BIAS_BYTES);
        __syncthreads();
    }
}

template <int C, bool SAVE>
int launch_qkv_attn(const QkvAttnArgs& a, hipStream_t st) {
    static std::atomic<unsigned long long> configured{0};
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&qkv_attn_fwd_kernel<C, SAVE>), QGeom<C>::LDS, configured)) return rc;
    long long items = (long long)a.nb * a.heads;
    int grid = items < 256 ? (int)items : 256 / a.heads * a.heads;        // a multiple of `heads`: workgroup b owns head b % heads
    hipLaunchKernelGGL((qkv_attn_fwd_kernel<C, SAVE>), dim3(grid), dim3(QTHREADS), QGeom<C>::LDS, st, a);
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
