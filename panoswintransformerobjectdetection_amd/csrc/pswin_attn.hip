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
//     the wave loops over the images of the batch that share that bias (window n = rep*nb + wb).  The bias
//     d(i,j) * alpha[idx(i,j)][h] + beta[idx(i,j)][h] (+ mask) is computed ONCE per work item from the cached
//     distance tile and the two 169-entry table columns (staged in LDS), stays in 64 VGPRs and enters every
//     score tile as the MFMA C operand: S + bias costs nothing per image.
//   * padded key columns get bias -inf, so padding needs no masking code.
//   * the next image's Q/K/V(/dO) fragments are requested from HBM before the current image's softmax / PV
//     (forward) resp. at the top of the iteration into a second register set (backward): one wave keeps
//     9-12 KB of loads in flight at all times.
//   * forward computes S^T = K.Q^T (key on the accumulator row, query on the lane): a score row lives in one
//     lane (+ its 3 partner lanes), softmax needs 2 cross-lane steps, and P is already the B operand of
//     O^T = V^T.P^T.  V^T comes from LDS through ds_read_b64_tr_b16 (bf16) / scalar LDS reads (f32).
//   * backward computes S = Q.K^T and dP = dO.V^T with the query on the accumulator row: P and dS are then
//     already the B operands of dV^T = dO^T.P and dK^T = Q^T.dS; only dS crosses LDS once (for dQ).
//     sum over the batch loop of dS (the bias gradient) stays in 64 VGPRs; at the end of the work item it is
//     binned into the 169 table entries (x 1 for beta, x d for alpha) with LDS atomics and written as one
//     1.4 KB partial per work item; a fixed-order column sum over the work items finishes dalpha / dbeta.
//   * dtype f32 uses the exact-f32 MFMA (v_mfma_f32_16x16x4_f32, k-ordered fmaf chain) so the f32 path
//     matches the PyTorch reference to summation order; dtype bf16 uses v_mfma_f32_16x16x32_bf16 with f32
//     softmax / accumulation.
//   * Q/K/V/dO fragments are loaded straight from HBM in MFMA operand layout (lane = row & 15, 8 contiguous
//     head-dim elements per lane = one 16-byte load for bf16): no staging pass for the row-wise operands.
#include <cstdlib>
#include <type_traits>

#include "pswin_attn_frag.hpp"

using namespace pswin;

namespace {

struct AttnArgs {
    const void *q, *k, *v;
    const float* dist;      // fwd: [n_dist][64 i][64 j] ; bwd: transposed tiles [n_dist][64 j][64 i] ; may be null
    const float* mask;      // same layouts, additive mask tiles ; may be null
    const float* alpha;     // [169][heads] (read only when dist != null)
    const float* beta;      // [169][heads]
    void* out;              // fwd
    const void* dout;       // bwd
    float* lse;             // fwd: written ; bwd: read
#ifdef PSWIN_ATTN_STAMPS
    unsigned long long* stamps;   // tools/probe/attn_probe.hip: per wave, 8 accumulated s_memtime intervals
#endif
    void *dq, *dk, *dv;     // bwd
    float* dtab;            // bwd: [n_items][64 j][64 i] sums of dS over the item's images, or null
    int n_dist, n_mask;
    int ld_qkv, ld_out, ld_dqkv;
    // (backward) where window w, head h of q / k / v starts, in elements from the q / k / v pointer: w * qkv_win_stride +
    // h * qkv_head_stride.  Row-major [n*49, ld_qkv] buffers: 49 * ld_qkv and 32; pswin_attn_bwd_ex takes any other packing.
    long long qkv_win_stride;
    int qkv_head_stride;
    int nb, heads, reps_per_chunk, n_items;
    float scale;
};


// stage column h of the [169][heads] tables into this wave's LDS: all loads first (tables_fetch), the LDS writes later
// (tables_store), so that they share one memory round trip with the other loads of a kernel's preamble
struct TabRaw {
    float b[3], a[3];
};
static_assert(NBINS <= 3 * 64, "three table entries per lane");
__device__ inline TabRaw tables_fetch(const AttnArgs& a, int h, int lane) {
    // buffer loads: entries past the table and the alpha column of a planar layer (no distance tiles) read as 0, no branches
    const __amdgpu_buffer_rsrc_t bres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.beta), 0, NBINS * a.heads * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.alpha), 0, a.dist ? NBINS * a.heads * 4 : 0, 0x00020000);
    TabRaw r;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int off = ((lane + 64 * k) * a.heads + h) * 4;
        r.b[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bres, off, 0, 0));
        r.a[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ares, off, 0, 0));
    }
    return r;
}
__device__ inline void tables_store(const TabRaw& r, int lane, float* tab_a, float* tab_b) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int t = lane + 64 * k;
        if (t < NBINS) {
            tab_b[t] = r.b[k];
            tab_a[t] = r.a[k];
        }
    }
}


template <int DT>
struct Tiles3 {
    Frag<DT> q[4], k[4], v[4];
};
template <int DT>
struct Tiles4 {
    Frag<DT> q[4], k[4], v[4], d[4];
};

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
// NQ = query tiles (of 16) per work item: 4 = the whole window, 2 = half of its queries (query rows are independent
// in the forward pass, so a window-head splits into two work items that both read all of K and V: half the bias /
// score registers per wave -> 3 waves per SIMD instead of 2, twice the items for load balance; K/V are re-read from L2).
template <int DT, int NQ>
struct FwdTiles {
    Frag<DT> q[NQ], k[4], v[4];
};

template <int DT, int WAVES, int NQ>
__global__ __launch_bounds__(64 * WAVES, (DT == PSWIN_BF16 ? (NQ == 2 ? 3 : 2) : 1)) void attn_fwd_kernel(AttnArgs a) {
    using VImg = LdsImg<DT, HD>;
    constexpr int WBYTES = VImg::BYTES + 2 * TABP * 4;
    constexpr int NSPLIT = 4 / NQ;
    __shared__ __attribute__((aligned(16))) char smem[WAVES * WBYTES];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // scalar item math
    const int c = lane & 15, g = lane >> 4;
    const int item = blockIdx.x * WAVES + wave;
    if (item >= a.n_items) return;   // wave-uniform
    const int qs = item % NSPLIT;    // which part of the queries
    const int rest = item / NSPLIT;
    const int h = rest % a.heads;
    const int wb = (rest / a.heads) % a.nb;
    const int chunk = rest / (a.heads * a.nb);
    const int ti0 = qs * NQ;
    char* vimg = smem + wave * WBYTES;
    float* tab_a = reinterpret_cast<float*>(vimg + VImg::BYTES);
    float* tab_b = tab_a + TABP;

    unsigned kv_off[4], q_off[NQ], o_off[NQ];          // per-lane byte offsets within a window, loop invariant
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) kv_off[tt] = row_off<DT>(16 * tt + c, a.ld_qkv, g);
#pragma unroll
    for (int tq = 0; tq < NQ; ++tq) {
        q_off[tq] = row_off<DT>(16 * (ti0 + tq) + c, a.ld_qkv, g);
        o_off[tq] = (unsigned)((16 * (ti0 + tq) + c) * a.ld_out) * ES<DT>;
    }
    auto load_tiles = [&](int r, FwdTiles<DT, NQ>& t) {
        const size_t row0 = ((size_t)(chunk * a.reps_per_chunk + r) * a.nb + wb) * TOK;
        const size_t head0 = (row0 * (size_t)a.ld_qkv + h * HD) * ES<DT>;
        const rsrc_t qb = window_rsrc<DT>(a.q, head0, a.ld_qkv);
        const rsrc_t kb = window_rsrc<DT>(a.k, head0, a.ld_qkv);
        const rsrc_t vb = window_rsrc<DT>(a.v, head0, a.ld_qkv);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            t.k[tt] = load_frag_at<DT>(kb, kv_off[tt]);
            t.v[tt] = load_frag_at<DT>(vb, kv_off[tt]);
        }
#pragma unroll
        for (int tq = 0; tq < NQ; ++tq) t.q[tq] = load_frag_at<DT>(qb, q_off[tq]);
    };
    FwdTiles<DT, NQ> cur;
    load_tiles(0, cur);              // in flight while the bias is being built

    const float inv_scale = 1.0f / a.scale;
    const float sl2e = a.scale * LOG2E;          // scores are kept unscaled: p = exp2((s' - m') * scale * log2 e)
    // bias of (wb, h) / scale for this item's queries: query on the lane, 4 consecutive keys per quad; in VGPRs for
    // the whole batch loop.  Its table columns and distance / mask quads are all requested before the first is used.
    f32x4 bias[NQ][4];
    {
        const float* dtile = a.dist ? a.dist + (size_t)(wb % a.n_dist) * (PADT * PADT) : nullptr;
        const float* mtile = a.mask ? a.mask + (size_t)(wb % a.n_mask) * (PADT * PADT) : nullptr;
        const TabRaw tabs = tables_fetch(a, h, lane);
        BiasRaw raw[NQ][4];
#pragma unroll
        for (int tq = 0; tq < NQ; ++tq)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) raw[tq][tj] = bias_fetch<false>(dtile, mtile, 16 * (ti0 + tq) + c, 16 * tj + 4 * g);
        __builtin_amdgcn_sched_barrier(0);
        tables_store(tabs, lane, tab_a, tab_b);
        __builtin_amdgcn_wave_barrier();
        // table lookups two quads at a time (all 8 at once would cost this kernel its third wave per SIMD)
#pragma unroll
        for (int tq = 0; tq < NQ; ++tq)
#pragma unroll
            for (int tj = 0; tj < 4; tj += 2) {
                const BiasTab t0 = bias_lookup<false>(tab_a, tab_b, 16 * (ti0 + tq) + c, 16 * tj + 4 * g);
                const BiasTab t1 = bias_lookup<false>(tab_a, tab_b, 16 * (ti0 + tq) + c, 16 * (tj + 1) + 4 * g);
                bias[tq][tj] = bias_from<false>(raw[tq][tj], t0, 16 * (ti0 + tq) + c, 16 * tj + 4 * g, inv_scale);
                bias[tq][tj + 1] = bias_from<false>(raw[tq][tj + 1], t1, 16 * (ti0 + tq) + c, 16 * (tj + 1) + 4 * g, inv_scale);
            }
    }

    for (int r = 0; r < a.reps_per_chunk; ++r) {
        const size_t win = (size_t)(chunk * a.reps_per_chunk + r) * a.nb + wb;
        const size_t row0 = win * TOK;
        const rsrc_t ob = window_rsrc<DT>(a.out, (row0 * (size_t)a.ld_out + h * HD) * ES<DT>, a.ld_out);
#pragma unroll
        for (int t = 0; t < 4; ++t) lds_write_frag<DT, HD>(vimg, 16 * t + c, 8 * g, cur.v[t]);   // rows >= 49: zeros
        // S^T tiles: keys 16 tj + 4 g + e on the accumulator rows, query 16 (ti0 + tq) + c on the lane
        f32x4 s4[NQ][4];
#pragma unroll
        for (int tq = 0; tq < NQ; ++tq)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) s4[tq][tj] = mma32<DT>(cur.k[tj], cur.q[tq], bias[tq][tj]);
        // the operand registers are free: request the next image now, its latency hides under softmax + PV
        if (r + 1 < a.reps_per_chunk) load_tiles(r + 1, cur);
        // V^T operand fragments: [k-step s][d tile dt]
        Frag<DT> vt[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) vt[s][dt] = lds_read_tr<DT, HD>(vimg, 32 * s + 4 * g, 16 * dt, c, g);

        // softmax in phases over the query tiles, so that the independent cross-lane chains overlap
        float m[NQ], l[NQ];
#pragma unroll
        for (int tq = 0; tq < NQ; ++tq) {
            float mm = s4[tq][3][0];                 // key tile 3 holds only key 48 (element 0 of group 0): rest is -inf
#pragma unroll
            for (int tj = 0; tj < 3; ++tj)
#pragma unroll
                for (int e = 0; e < 4; ++e) mm = fmaxf(mm, s4[tq][tj][e]);
            m[tq] = mm;
        }
#pragma unroll
        for (int tq = 0; tq < NQ; ++tq) m[tq] = group_max(m[tq]);
#pragma unroll
        for (int tq = 0; tq < NQ; ++tq) {
            const float mb = -m[tq] * sl2e;
            float ll = 0.f;
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (tj == 3 && e > 0) {
                        s4[tq][tj][e] = 0.f;         // keys 49..63: bias -inf, weight exactly 0
                    } else {
                        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s4[tq][tj][e], sl2e, mb));
                        s4[tq][tj][e] = p;
                        ll += p;
                    }
                }
            l[tq] = ll;
        }
#pragma unroll
        for (int tq = 0; tq < NQ; ++tq) l[tq] = group_sum(l[tq]);
        // O^T[d][i] = sum_j V^T[d][j] P^T[j][i]
#pragma unroll
        for (int tq = 0; tq < NQ; ++tq) {
            f32x4 o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const Frag<DT> pf = pack_frag<DT>(s4[tq][2 * s], s4[tq][2 * s + 1]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) o[dt] = mma32<DT>(vt[s][dt], pf, o[dt]);
            }
            const int i = 16 * (ti0 + tq) + c;
            const float inv_l = 1.0f / l[tq];
            // (the lane exchange inside store_row8 needs all lanes: rows >= 49 only skip the store itself)
            {
                const f32x4 o0 = o[0] * inv_l, o1 = o[1] * inv_l;
                store_row8_at<DT>(ob, o_off[tq], g, o0, o1);
            }
            if (g == 0) {
                float* lse_row = a.lse + (win * a.heads + h) * PADT;
                lse_row[(unsigned)i] = (i < TOK) ? __builtin_fmaf(m[tq], a.scale, logf(l[tq])) : INFINITY;
            }
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
    static constexpr int BYTES = 3 * ROW + TT + 2 * TABP * 4;   // + table columns (alpha, beta)
};

template <int DT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void attn_bwd_kernel(AttnArgs a) {
    using L = BwdLds<DT>;
    __shared__ __attribute__((aligned(16))) char smem[WAVES * L::BYTES];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
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
    float* tab_a = reinterpret_cast<float*>(timg + L::TT);
    float* tab_b = tab_a + TABP;
    const float* dtile = a.dist ? a.dist + (size_t)(wb % a.n_dist) * (PADT * PADT) : nullptr;   // transposed: [j][i]
    const float* mtile = a.mask ? a.mask + (size_t)(wb % a.n_mask) * (PADT * PADT) : nullptr;

    unsigned in_off[4], do_off[4], dx_off[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        in_off[tt] = row_off<DT>(16 * tt + c, a.ld_qkv, g);
        do_off[tt] = row_off<DT>(16 * tt + c, a.ld_out, g);
        dx_off[tt] = (unsigned)((16 * tt + c) * a.ld_dqkv) * ES<DT>;
    }
    auto load_tiles = [&](int r, Tiles4<DT>& t) {
        const size_t win_ = (size_t)(chunk * a.reps_per_chunk + r) * a.nb + wb, row0 = win_ * TOK;
        const size_t head0 = (win_ * (size_t)a.qkv_win_stride + (size_t)h * a.qkv_head_stride) * ES<DT>;
        const rsrc_t qb = window_rsrc<DT>(a.q, head0, a.ld_qkv);
        const rsrc_t kb = window_rsrc<DT>(a.k, head0, a.ld_qkv);
        const rsrc_t vb = window_rsrc<DT>(a.v, head0, a.ld_qkv);
        const rsrc_t db = window_rsrc<DT>(a.dout, (row0 * (size_t)a.ld_out + h * HD) * ES<DT>, a.ld_out);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            t.q[tt] = load_frag_at<DT>(qb, in_off[tt]);
            t.k[tt] = load_frag_at<DT>(kb, in_off[tt]);
            t.v[tt] = load_frag_at<DT>(vb, in_off[tt]);
            t.d[tt] = load_frag_at<DT>(db, do_off[tt]);
        }
    };

    constexpr bool DUAL = false;   // (DT == PSWIN_BF16) measured slower: the extra 64 live registers turn into AGPR<->VGPR copies
    Tiles4<DT> cur, nxt;
    load_tiles(0, cur);              // in flight while the bias is being built

    const float inv_scale = 1.0f / a.scale;
    const float sl2e = a.scale * LOG2E;
    // bias / scale in the backward layout: rows i = 16 ti + 4 g + e on the registers, key j = 16 tj + c on the lane
    f32x4 bias[4][4];
    {
        const TabRaw tabs = tables_fetch(a, h, lane);
        BiasRaw raw[4][4];
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) raw[ti][tj] = bias_fetch<true>(dtile, mtile, 16 * ti + 4 * g, 16 * tj + c);
        __builtin_amdgcn_sched_barrier(0);
        tables_store(tabs, lane, tab_a, tab_b);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
                bias[ti][tj] = bias_from<true>(raw[ti][tj], bias_lookup<true>(tab_a, tab_b, 16 * ti + 4 * g, 16 * tj + c), 16 * ti + 4 * g, 16 * tj + c,
                                               inv_scale);
    }

    f32x4 gsum[4][4];   // sum over the batch loop of dS
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) gsum[ti][tj] = f32x4{0.f, 0.f, 0.f, 0.f};

    // bf16: a second register set holds the next image's operands for a whole iteration (64 VGPRs).  f32 operands
    // are twice as large, so the f32 (parity) path reuses the one set as soon as the MFMAs have consumed it.
    for (int r = 0; r < a.reps_per_chunk; ++r) {
        const size_t win = (size_t)(chunk * a.reps_per_chunk + r) * a.nb + wb;
        const size_t row0 = win * TOK;
        if constexpr (DUAL) {
            if (r + 1 < a.reps_per_chunk) load_tiles(r + 1, nxt);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            lds_write_frag<DT, HD>(qimg, 16 * t + c, 8 * g, cur.q[t]);
            lds_write_frag<DT, HD>(kimg, 16 * t + c, 8 * g, cur.k[t]);
            lds_write_frag<DT, HD>(doimg, 16 * t + c, 8 * g, cur.d[t]);
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
                f32x4 lse4 = *reinterpret_cast<const f32x4*>(lse_row + 16 * ti + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) lse4[e] = -lse4[e] * LOG2E;       // p = exp2(s' * scale*log2e - lse*log2e)
                f32x4 delta = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) {
                    const f32x4 sc = mma32<DT>(cur.q[ti], cur.k[tj], bias[ti][tj]);                  // (S + bias) / scale
                    const f32x4 dp = mma32<DT>(cur.d[ti], cur.v[tj], f32x4{0.f, 0.f, 0.f, 0.f});     // dP[i][j]
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // query tile 3 holds only query 48 (row e = 0 of group 0); its other rows have lse = +inf
                        const float p = (ti == 3 && e > 0) ? 0.f : __builtin_amdgcn_exp2f(__builtin_fmaf(sc[e], sl2e, lse4[e]));
                        p4[tt][tj][e] = p;
                        delta[e] = __builtin_fmaf(p, dp[e], delta[e]);
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
                const Frag<DT> pf = pack_frag<DT>(p4[0][tj], p4[1][tj]);
                const Frag<DT> dsf = pack_frag<DT>(ds4[0][tj], ds4[1][tj]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt][tj] = mma32<DT>(dot[dt], pf, dv[dt][tj]);
                    dk[dt][tj] = mma32<DT>(qt[dt], dsf, dk[dt][tj]);
                }
            }
        }
        if constexpr (!DUAL) {
            if (r + 1 < a.reps_per_chunk) load_tiles(r + 1, cur);
        }
        const size_t dhead0 = (row0 * (size_t)a.ld_dqkv + h * HD) * ES<DT>;
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
            store_row8_at<DT>(window_rsrc<DT>(a.dv, dhead0, a.ld_dqkv), dx_off[tj], g, dv[0][tj], dv[1][tj]);
            store_row8_at<DT>(window_rsrc<DT>(a.dk, dhead0, a.ld_dqkv), dx_off[tj], g, dk[0][tj] * a.scale, dk[1][tj] * a.scale);
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
                const Frag<DT> tf = lds_read_tr<DT, PADT>(timg, 32 * s + 4 * g, 16 * ti, c, g);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) dq[dt][ti] = mma32<DT>(kt[dt], tf, dq[dt][ti]);
            }
        }
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            store_row8_at<DT>(window_rsrc<DT>(a.dq, dhead0, a.ld_dqkv), dx_off[ti], g, dq[0][ti] * a.scale, dq[1][ti] * a.scale);
        }
        if constexpr (DUAL) {
            if (r + 1 < a.reps_per_chunk) cur = nxt;
        }
    }
    // sum over this work item's images of dS, as a transposed [key j][query i] tile; binned into the tables by
    // dtab_partial_kernel (LDS float atomics in here cost ~100 cycles per wave-instruction: measured 1.6x the kernel)
    if (a.dtab) {
        float* gt = a.dtab + (size_t)item * (PADT * PADT);
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
                *reinterpret_cast<f32x4*>(gt + (16 * tj + c) * PADT + 16 * ti + 4 * g) = gsum[ti][tj];
    }
}

// ---------------------------------------------------------------------------------------------
// backward, two waves per (window, head) work item (bf16)
// ---------------------------------------------------------------------------------------------
// The single-wave backward above needs ~450 registers (bias 64 + dS-sum 64 + dV/dK accumulators 64 + operands 64 + P/dS
// 64 + ...), i.e. one wave per SIMD: every LDS round trip, MFMA result latency and memory wait is exposed (measured
// ~19 cycles per instruction).  Here a workgroup = 2 waves = one work item, split by QUERY halves (wave w owns query
// tiles 2w, 2w+1): bias, dS-sum, P and dS are halved, delta and dQ are complete per wave (they contract over keys, and
// each wave sees all keys); only dV / dK contract over queries, so each wave holds a partial over its 32 queries and
// the two exchange ONE partial each through LDS (wave 0 finishes dV, wave 1 finishes dK).  ~200 registers -> two waves
// per SIMD.  K and V are loaded by both waves (second read hits L2).
struct PairLds {
    static constexpr int KIMG = 64 * 64;            // K image [64 keys][32] bf16 (shared: transposed reads for dQ), img32_off
    static constexpr int HALF = 32 * 64;            // per wave: Q / dO images of its 32 query rows, img32_off
    static constexpr int TIMG = 64 * 64;            // per wave: dS^T [64 keys][32 own queries], timg_off
    static constexpr int EXCH = 64 * 128;           // one f32 partial [64 keys][32 d], exch_off
    static constexpr int TAB = 2 * TABP * 4;
    static constexpr int BYTES = KIMG + 2 * (2 * HALF + TIMG) + 2 * EXCH + TAB;
};

// dS^T image: 64-byte rows of eight 8-byte units, written by columns (ds_write_b64: 16 rows, one unit) and read
// transposed (8 rows x 4 units per half-wave).  Unit XOR (row bits 1, 3, 2): rows of equal parity within 16 get 8
// different units (writes conflict-free), rows r, r + 4 differ in unit bit 2 (reads conflict-free).  Unchanged by +16.
__device__ inline int timg_off(int row, int unit8) {
    return row * 64 + ((unit8 ^ (((row >> 1) & 1) | (((row >> 3) & 1) << 1) | (((row >> 2) & 1) << 2))) << 3);
}
// f32 exchange rows: 128 bytes = eight 16-byte chunks, XOR row & 7 (8 rows of one ds_*_b128 group -> 8 different chunks)
__device__ inline int exch_off(int row, int chunk16) { return row * 128 + ((chunk16 ^ (row & 7)) << 4); }

// transposed read of a 32-column bf16 image with 32-row operand blocks (rows R0+{0..3}, R0+16+{0..3})
template <bool TIMG>
__device__ inline Frag<PSWIN_BF16> tr32(const char* img, int R0, int col0, int c) {
    const int q = c >> 2, p = c & 3;
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
    const int k8 = (col0 >> 2) + p;
    const char* a0 = img + (TIMG ? timg_off(R0 + q, k8) : img32_off(R0 + q, k8 >> 1) + ((k8 & 1) << 3));
    const char* a1 = a0 + 16 * 64;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    Frag<PSWIN_BF16> f;
    f.v = __builtin_bit_cast(bf16x8, both);
    return f;
}

#ifdef PSWIN_ATTN_STAMPS
#define PSWIN_STAMP(i)                                                  \
    do {                                                                \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();     \
        stamp_acc[i] += t_ - stamp_last;                                \
        stamp_last = t_;                                                \
    } while (0)
#else
#define PSWIN_STAMP(i)
#endif
__global__ __launch_bounds__(128, 2) void attn_bwd_pair_kernel(AttnArgs a) {
    constexpr int DT = PSWIN_BF16;
#ifdef PSWIN_ATTN_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
#endif
    __shared__ __attribute__((aligned(16))) char smem[PairLds::BYTES];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;      // w = query half
    const int c = lane & 15, g = lane >> 4;
    const int item = blockIdx.x;
    const int h = item % a.heads;
    const int wb = (item / a.heads) % a.nb;
    const int chunk = item / (a.heads * a.nb);
    char* kimg = smem;
    char* qimg = kimg + PairLds::KIMG + w * (2 * PairLds::HALF + PairLds::TIMG);
    char* doimg = qimg + PairLds::HALF;
    char* timg = doimg + PairLds::HALF;
    float* exch_dk = reinterpret_cast<float*>(smem + PairLds::KIMG + 2 * (2 * PairLds::HALF + PairLds::TIMG));   // from wave 0
    float* exch_dv = exch_dk + 64 * 32;                                                                        // from wave 1
    float* tab_a = exch_dv + 64 * 32;
    float* tab_b = tab_a + TABP;
    const float* dtile = a.dist ? a.dist + (size_t)(wb % a.n_dist) * (PADT * PADT) : nullptr;   // transposed: [j][i]
    const float* mtile = a.mask ? a.mask + (size_t)(wb % a.n_mask) * (PADT * PADT) : nullptr;

    struct Ops {
        Frag<DT> q[2], d[2], k[4], v[4];
        float lse;                    // log-sum-exp of query 32 w + (lane & 31)
    };
    unsigned kv_off[4], q_off[2], do_off[2], dkv_off[4], dq_off[2];   // per-lane byte offsets within a window
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        kv_off[tt] = row_off<DT>(16 * tt + c, a.ld_qkv, g);
        dkv_off[tt] = (unsigned)((16 * tt + c) * a.ld_dqkv) * 2u;
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        q_off[tt] = row_off<DT>(16 * (2 * w + tt) + c, a.ld_qkv, g);
        do_off[tt] = row_off<DT>(16 * (2 * w + tt) + c, a.ld_out, g);
        dq_off[tt] = (unsigned)((16 * (2 * w + tt) + c) * a.ld_dqkv) * 2u;
    }
    auto load_ops = [&](int r, Ops& t) {
        const size_t win_ = (size_t)(chunk * a.reps_per_chunk + r) * a.nb + wb, row0 = win_ * TOK;
        const size_t head0 = (win_ * (size_t)a.qkv_win_stride + (size_t)h * a.qkv_head_stride) * 2;
        const rsrc_t qb = window_rsrc<DT>(a.q, head0, a.ld_qkv);
        const rsrc_t kb = window_rsrc<DT>(a.k, head0, a.ld_qkv);
        const rsrc_t vb = window_rsrc<DT>(a.v, head0, a.ld_qkv);
        const rsrc_t db = window_rsrc<DT>(a.dout, (row0 * (size_t)a.ld_out + h * HD) * 2, a.ld_out);
        // The log-sum-exp values of this wave's 32 queries travel with the operands, one per lane, requested FIRST (loads return
        // in order: they are there whenever an operand fragment is) and are handed to the lanes that need them by ds_bpermute.
        // Loaded where they are used, at the top of the next iteration, they were a memory round trip per image that nothing
        // covered: that load sits behind the previous image's stores in the one in-order counter.
        t.lse = a.lse[(win_ * a.heads + h) * PADT + 32 * w + (lane & 31)];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            t.k[tt] = load_frag_at<DT>(kb, kv_off[tt]);
            t.v[tt] = load_frag_at<DT>(vb, kv_off[tt]);
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            t.q[tt] = load_frag_at<DT>(qb, q_off[tt]);
            t.d[tt] = load_frag_at<DT>(db, do_off[tt]);
        }
    };
    Ops cur;
    load_ops(0, cur);                 // in flight while the bias is being built

    const float inv_scale = 1.0f / a.scale;
    const float sl2e = a.scale * LOG2E;
    // bias / scale for this wave's query tiles: rows i = 16 (2w + tt) + 4 g + e on the registers, key j = 16 tj + c.  The table
    // columns and the distance / mask quads are all requested before the first is used (one round trip, not nine).
    f32x4 bias[2][4], gsum[2][4];
    {
        TabRaw tabs;
        if (w == 0) tabs = tables_fetch(a, h, lane);
        BiasRaw raw[2][4];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) raw[tt][tj] = bias_fetch<true>(dtile, mtile, 16 * (2 * w + tt) + 4 * g, 16 * tj + c);
        __builtin_amdgcn_sched_barrier(0);
        if (w == 0) tables_store(tabs, lane, tab_a, tab_b);
        __syncthreads();
        BiasTab tab[2][4];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) tab[tt][tj] = bias_lookup<true>(tab_a, tab_b, 16 * (2 * w + tt) + 4 * g, 16 * tj + c);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) {
                bias[tt][tj] = bias_from<true>(raw[tt][tj], tab[tt][tj], 16 * (2 * w + tt) + 4 * g, 16 * tj + c, inv_scale);
                gsum[tt][tj] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    }

    PSWIN_STAMP(0);
    for (int r = 0; r < a.reps_per_chunk; ++r) {
        const size_t win = (size_t)(chunk * a.reps_per_chunk + r) * a.nb + wb;
        const size_t row0 = win * TOK;
        const size_t dhead0 = (row0 * (size_t)a.ld_dqkv + h * HD) * 2;
        // LDS images for the transposed operand reads
        if (w == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) *reinterpret_cast<bf16x8*>(kimg + img32_off(16 * t + c, g)) = cur.k[t].v;
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            *reinterpret_cast<bf16x8*>(qimg + img32_off(16 * tt + c, g)) = cur.q[tt].v;
            *reinterpret_cast<bf16x8*>(doimg + img32_off(16 * tt + c, g)) = cur.d[tt].v;
        }
        Frag<DT> pf[4], dsf[4];
        {
            f32x4 p4[2][4], ds4[2][4];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                f32x4 lse4;               // rows 16 tt + 4 g + e of this wave's half
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    lse4[e] = -LOG2E * __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (16 * tt + 4 * g + e), __builtin_bit_cast(int, cur.lse)));
                f32x4 delta = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) {
                    const f32x4 sc = mma32<DT>(cur.q[tt], cur.k[tj], bias[tt][tj]);                  // (S + bias) / scale
                    const f32x4 dp = mma32<DT>(cur.d[tt], cur.v[tj], f32x4{0.f, 0.f, 0.f, 0.f});     // dP[i][j]
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[e], sl2e, lse4[e]));   // rows >= 49: lse = +inf -> 0
                        p4[tt][tj][e] = p;
                        delta[e] = __builtin_fmaf(p, dp[e], delta[e]);
                    }
                    ds4[tt][tj] = dp;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) delta[e] = row16_sum(delta[e]);
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) ds4[tt][tj][e] = p4[tt][tj][e] * (ds4[tt][tj][e] - delta[e]);
                    gsum[tt][tj] = gsum[tt][tj] + ds4[tt][tj];
                }
            }
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) {
                pf[tj] = pack_frag<DT>(p4[0][tj], p4[1][tj]);
                dsf[tj] = pack_frag<DT>(ds4[0][tj], ds4[1][tj]);
                // dS^T[j][local i .. +3], local i = 16 tt + 4 g: the two halves of the fragment just packed (one rounding, one
                // conversion per value)
                typedef __attribute__((ext_vector_type(2))) unsigned long long u64x2_t;
                const u64x2_t halves = __builtin_bit_cast(u64x2_t, dsf[tj].v);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
                    *reinterpret_cast<unsigned long long*>(timg + timg_off(16 * tj + c, 4 * tt + g)) = halves[tt];
            }
        }
        PSWIN_STAMP(1);
        // operands consumed: request the next image
        if (r + 1 < a.reps_per_chunk) load_ops(r + 1, cur);

        // partial dV^T / dK^T over this wave's 32 queries: [dt][tj], rows d = 16 dt + 4 g + e, column key j = 16 tj + c.
        // wave 0 keeps dV (= dO^T P) and sends dK (= Q^T dS), wave 1 keeps dK and sends dV.  The role is applied to the OPERANDS
        // (which LDS image the transposed fragments come from: a scalar select; which packed fragment multiplies them: 32
        // register selects) instead of to the 64 result registers of both products.
        const char* keep_img = w == 0 ? doimg : qimg;
        const char* send_img = w == 0 ? qimg : doimg;
        Frag<DT> ka[2], sa[2], kb[4], sb[4];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            ka[dt] = tr32<false>(keep_img, 4 * g, 16 * dt, c);
            sa[dt] = tr32<false>(send_img, 4 * g, 16 * dt, c);
        }
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
            kb[tj] = w == 0 ? pf[tj] : dsf[tj];
            sb[tj] = w == 0 ? dsf[tj] : pf[tj];
        }
        f32x4 keep[2][4];
        {
            float* exch = w == 0 ? exch_dk : exch_dv;
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    keep[dt][tj] = mma32<DT>(ka[dt], kb[tj], z);
                    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(exch) + exch_off(16 * tj + c, 4 * dt + g)) = mma32<DT>(sa[dt], sb[tj], z);
                }
        }
        PSWIN_STAMP(2);
        __syncthreads();          // exchange written, K image written
        PSWIN_STAMP(3);
        {
            const float* other = w == 0 ? exch_dv : exch_dk;
            const rsrc_t dst = window_rsrc<DT>(w == 0 ? a.dv : a.dk, dhead0, a.ld_dqkv);
            const float mul = w == 0 ? 1.0f : a.scale;
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) {
                const int j = 16 * tj + c;
                const char* ob = reinterpret_cast<const char*>(other);
                const f32x4 o0 = (keep[0][tj] + *reinterpret_cast<const f32x4*>(ob + exch_off(j, g))) * mul;
                const f32x4 o1 = (keep[1][tj] + *reinterpret_cast<const f32x4*>(ob + exch_off(j, 4 + g))) * mul;
                store_row8_at<DT>(dst, dkv_off[tj], g, o0, o1);
            }
        }
        PSWIN_STAMP(4);
        // dQ^T[d][i] = scale * sum_j K^T[d][j] dS^T[j][i] for this wave's queries
        f32x4 dq[2][2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) dq[dt][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            Frag<DT> kt[2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) kt[dt] = tr32<false>(kimg, 32 * s + 4 * g, 16 * dt, c);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const Frag<DT> tf = tr32<true>(timg, 32 * s + 4 * g, 16 * tt, c);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) dq[dt][tt] = mma32<DT>(kt[dt], tf, dq[dt][tt]);
            }
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            store_row8_at<DT>(window_rsrc<DT>(a.dq, dhead0, a.ld_dqkv), dq_off[tt], g, dq[0][tt] * a.scale, dq[1][tt] * a.scale);
        }
        PSWIN_STAMP(5);
        __syncthreads();          // both waves are done with the K image and the exchange buffers
        PSWIN_STAMP(6);
    }
#ifdef PSWIN_ATTN_STAMPS
    if (a.stamps && lane == 0)
        for (int i = 0; i < 8; ++i) a.stamps[((size_t)blockIdx.x * 2 + w) * 8 + i] = stamp_acc[i];
#endif
    if (a.dtab) {
        float* gt = a.dtab + (size_t)item * (PADT * PADT);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
                *reinterpret_cast<f32x4*>(gt + (16 * tj + c) * PADT + 16 * (2 * w + tt) + 4 * g) = gsum[tt][tj];
    }
}

// ---------------------------------------------------------------------------------------------
// helpers around the main kernels
// ---------------------------------------------------------------------------------------------
// [n][49][49] -> [n][64][64], optionally transposed, zero padded: the tile layout the attention kernels read
__global__ void pad_tiles_kernel(const float* __restrict__ src, int n, int transpose, float* __restrict__ dst) {
    const size_t t = blockIdx.x;
    for (int e = threadIdx.x; e < PADT * PADT; e += blockDim.x) {
        const int r = e / PADT, col = e - r * PADT;
        float v = 0.f;
        if (r < TOK && col < TOK) v = transpose ? src[(t * TOK + col) * TOK + r] : src[(t * TOK + r) * TOK + col];
        dst[t * PADT * PADT + e] = v;
    }
}

constexpr int DTAB_BLOCKS = 128;
// floats per partial row [heads][2][49*49], rounded up to 4 so that rows stay 16-byte aligned (the row sum of the
// partials can then run as a pswin_reduce_jobs job together with the other reductions of a backward pass)
__host__ __device__ inline int dtab_ld(int heads) { return (heads * 2 * TOK * TOK + 3) & ~3; }
// stage-1 workgroups per head: enough (blocks x heads >= ~500) to fill the chip, few enough to keep the partials small
inline int dtab_blocks(int n_tiles, int heads) {
    int nb = (512 + heads - 1) / heads;
    if (nb > DTAB_BLOCKS) nb = DTAB_BLOCKS;
    if (nb < 16) nb = 16;
    return n_tiles < nb ? n_tiles : nb;
}
constexpr int DTAB_THREADS = 256;

// Table-gradient jobs of one backward pass (one per attention module) run as TWO launches for all of them; the job table
// travels in the kernel arguments.
constexpr int TJ_MAX = 32;
struct TJob {
    const float* g;        // dScore sums [n_tiles][heads][64 j][64 i]
    const float* dist_t;   // [n_dist][64 j][64 i] or NULL
    float* partial;        // [blocks][heads][2][49*49]
    float* summed;         // [heads][2][49*49]: fixed-order sum of the block partials (colsum_kernel between the stages)
    float* dalpha;         // [169][heads] or NULL
    float* dbeta;          // [169][heads]
    int n_tiles, nb, n_dist, heads, blocks;
    int first1, first2;    // first workgroup of this job in the stage-1 / stage-2 grid
};
struct TBatch {
    TJob job[TJ_MAX];
    int n;
};
static_assert(sizeof(TBatch) <= 4000, "the job table must fit the kernel-argument segment");

// Stage 1: per (block, head) of a job, sum over a strided subset of the tiles of g and g * d for every (i, j) pair.  A
// thread owns the same pairs for every tile, so the sums stay in registers (no atomics).
__global__ __launch_bounds__(DTAB_THREADS) void dtab_partial_kernel(const TBatch b) {
    int jn = 0;
    for (int q = 1; q < b.n; ++q)
        if (b.job[q].first1 <= (int)blockIdx.x) jn = q;
    const float* __restrict__ g = b.job[jn].g;
    const float* __restrict__ dist_t = b.job[jn].dist_t;
    const int heads = b.job[jn].heads, n_tiles = b.job[jn].n_tiles, nb = b.job[jn].nb, n_dist = b.job[jn].n_dist;
    const int nblk = b.job[jn].blocks;
    const int local = (int)blockIdx.x - b.job[jn].first1;
    const int blk = local / heads, h = local - blk * heads;
    // 16-byte loads: a thread owns up to 3 quads (j, 4 i4 .. 4 i4 + 3) of the 49 x 13 quads that hold real pairs (the
    // 64 x 64 tiles are zero beyond 49 in both directions)
    constexpr int QPR = (TOK + 3) / 4, NQ = TOK * QPR, QPT = (NQ + DTAB_THREADS - 1) / DTAB_THREADS;
    f32x4 sb[QPT], sa[QPT];
    int qoff[QPT];
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
        sb[k] = sa[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int q = threadIdx.x + k * DTAB_THREADS;
        qoff[k] = q < NQ ? (q / QPR) * PADT + 4 * (q % QPR) : -1;
    }
    for (int tile = blk; tile < n_tiles; tile += nblk) {
        const float* gt = g + ((size_t)tile * heads + h) * (PADT * PADT);
        const float* dt = dist_t ? dist_t + (size_t)((tile % nb) % n_dist) * (PADT * PADT) : nullptr;
#pragma unroll
        for (int k = 0; k < QPT; ++k) {
            if (qoff[k] >= 0) {
                const f32x4 gval = *reinterpret_cast<const f32x4*>(gt + qoff[k]);
                sb[k] = sb[k] + gval;
                if (dt) sa[k] = sa[k] + gval * *reinterpret_cast<const f32x4*>(dt + qoff[k]);
            }
        }
    }
    float* out = b.job[jn].partial + (size_t)blk * dtab_ld(heads) + (size_t)h * 2 * TOK * TOK;
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
        if (qoff[k] >= 0) {
            const int j = qoff[k] / PADT, i0 = qoff[k] - j * PADT;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (i0 + e < TOK) {
                    out[j * TOK + i0 + e] = sb[k][e];
                    out[TOK * TOK + j * TOK + i0 + e] = sa[k][e];
                }
            }
        }
    }
}

// Stage 2 (after the fixed-order, coalesced column sum over the stage-1 blocks): one WAVE per (table entry, head) of a
// job: lane l < 49 is the key position (hj, wj) = (l / 7, l % 7) and loads the one pair (i, j) with hi - hj = dh,
// wi - wj = dw if it exists; a butterfly reduction (fixed order) finishes the bin.  (Summing the block partials in
// here instead, 128 strided 4-byte loads per lane, measured 32 us per module against 10 + 5 us.)
__global__ __launch_bounds__(256) void dtab_final_kernel(const TBatch b) {
    int jn = 0;
    for (int q = 1; q < b.n; ++q)
        if (b.job[q].first2 <= (int)blockIdx.x) jn = q;
    const int heads = b.job[jn].heads;
    const bool has_dist = b.job[jn].dist_t != nullptr;
    const float* __restrict__ summed = b.job[jn].summed;
    const int t = ((int)blockIdx.x - b.job[jn].first2) * 4 + (threadIdx.x >> 6);     // over 169 * heads, wave-uniform
    const int lane = threadIdx.x & 63;
    if (t >= NBINS * heads) return;
    const int idx = t / heads, h = t - idx * heads;
    const int dh = idx / (2 * PSWIN_WS - 1) - (PSWIN_WS - 1), dw = idx % (2 * PSWIN_WS - 1) - (PSWIN_WS - 1);
    const int hj = lane / PSWIN_WS, wj = lane - hj * PSWIN_WS;
    const int hi = hj + dh, wi = wj + dw;
    float sb = 0.f, sa = 0.f;
    if (lane < TOK && hi >= 0 && hi < PSWIN_WS && wi >= 0 && wi < PSWIN_WS) {
        const int e = lane * TOK + hi * PSWIN_WS + wi;     // j * 49 + i
        const float* p = summed + (size_t)h * 2 * TOK * TOK + e;
        sb = p[0];
        if (has_dist) sa = p[TOK * TOK];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        sb += __shfl_xor(sb, m, 64);
        sa += __shfl_xor(sa, m, 64);
    }
    if (lane == 0) {
        b.job[jn].dbeta[t] = sb;
        if (has_dist) b.job[jn].dalpha[t] = sa;
    }
}

// Work items per bias window.  More chunks = more independent waves but the per-item setup (bias build, and the dS-sum
// tile in the backward pass) is repeated and the batch loop that amortises it gets shorter.  Targets fitted to the
// PanoSwin-T stage shapes on MI355X (tools/bench_attn.py): forward (2 waves/SIMD) ~1100 items, backward
// (1 wave/SIMD, heavier items) ~600 items.
inline int pick_chunks(int reps, int nb, int heads, bool backward) {
    const long long target = backward ? 600 : 1100;          // swept inside the full step in round 2 (profiles/r02_*)
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
                             int heads, int dtype, const float* dist, int n_dist, const float* alpha,
                             const float* beta, const float* mask, int n_mask) {
    PSWIN_CHECK_ARG(q && k && v && beta);
    PSWIN_CHECK_ARG(valid_dtype(dtype));
    PSWIN_CHECK_ARG(n_windows > 0 && nb > 0 && heads > 0 && n_windows % nb == 0);
    PSWIN_CHECK_ARG(ld_qkv >= heads * HD && ld_qkv % 8 == 0);
    PSWIN_CHECK_ARG(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(dist) && aligned16(mask));
    PSWIN_CHECK_ARG((long long)n_windows * TOK * ld_qkv < (1ll << 40));
    PSWIN_CHECK_ARG(!dist || (alpha && n_dist > 0 && nb % n_dist == 0));
    PSWIN_CHECK_ARG(!mask || (n_mask > 0 && nb % n_mask == 0));
    return PSWIN_OK;
}

}  // namespace

extern "C" int pswin_attn_pad_tiles(const float* src, int n, int transpose, float* dst, void* stream) {
    PSWIN_CHECK_ARG(src && dst && n > 0 && aligned16(dst));
    hipLaunchKernelGGL(pad_tiles_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, src, n, transpose, dst);
    PSWIN_LAUNCH_RET();
}

extern "C" int pswin_attn_fwd(const void* q, const void* k, const void* v, int ld_qkv, const float* dist, int n_dist,
                              const float* alpha, const float* beta, const float* mask, int n_mask, void* out,
                              int ld_out, float* lse, int n_chunks, int n_windows, int n_bias_windows, int heads,
                              float scale, int dtype, void* stream) {
    int rc = check_attn_common(q, k, v, ld_qkv, n_windows, n_bias_windows, heads, dtype, dist, n_dist, alpha, beta,
                               mask, n_mask);
    if (rc) return rc;
    PSWIN_CHECK_ARG(out && lse && aligned16(out));
    PSWIN_CHECK_ARG(ld_out >= heads * HD && ld_out % 8 == 0);
    const int reps = n_windows / n_bias_windows;
    PSWIN_CHECK_ARG(n_chunks >= 0 && n_chunks <= reps && (n_chunks == 0 || reps % n_chunks == 0));
    const int chunks = n_chunks ? n_chunks : pick_chunks(reps, n_bias_windows, heads, false);
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v; a.dist = dist; a.mask = mask; a.alpha = alpha; a.beta = beta;
    a.n_dist = dist ? n_dist : 1; a.n_mask = mask ? n_mask : 1;
    a.out = out; a.lse = lse;
    a.ld_qkv = ld_qkv; a.ld_out = ld_out;
    a.nb = n_bias_windows; a.heads = heads; a.reps_per_chunk = reps / chunks;
    a.scale = scale;
    constexpr int W = 4;
    if (dtype == PSWIN_BF16) {
        a.n_items = chunks * n_bias_windows * heads * 2;         // two work items (query halves) per window-head
        hipLaunchKernelGGL((attn_fwd_kernel<PSWIN_BF16, W, 2>), dim3((a.n_items + W - 1) / W), dim3(64 * W), 0,
                           (hipStream_t)stream, a);
    } else {
        a.n_items = chunks * n_bias_windows * heads;
        hipLaunchKernelGGL((attn_fwd_kernel<PSWIN_F32, W, 4>), dim3((a.n_items + W - 1) / W), dim3(64 * W), 0,
                           (hipStream_t)stream, a);
    }
    PSWIN_LAUNCH_RET();
}

extern "C" int pswin_attn_suggest_chunks(int n_windows, int n_bias_windows, int heads, int backward) {
    if (n_windows <= 0 || n_bias_windows <= 0 || heads <= 0 || n_windows % n_bias_windows) return PSWIN_ERR_ARG;
    return pick_chunks(n_windows / n_bias_windows, n_bias_windows, heads, backward != 0);
}

extern "C" int pswin_attn_table_grads_partial_rows(int n_tiles, int heads) {
    return (n_tiles > 0 && heads > 0) ? dtab_blocks(n_tiles, heads) : PSWIN_ERR_ARG;
}

extern "C" int pswin_attn_table_grads_workspace(int heads) {
    return heads > 0 ? (DTAB_BLOCKS + 1) * dtab_ld(heads) : PSWIN_ERR_ARG;   // block partial rows + their sum
}

#ifdef PSWIN_ATTN_STAMPS
static unsigned long long* g_attn_stamps = nullptr;
extern "C" void pswin_attn_debug_stamps(unsigned long long* p) { g_attn_stamps = p; }
#endif

extern "C" int pswin_attn_bwd_ex(const void* q, const void* k, const void* v, int ld_qkv, long long qkv_window_stride,
                                 int qkv_head_stride, const float* dist_t, int n_dist, const float* alpha, const float* beta,
                                 const float* mask_t, int n_mask, const void* dout, int ld_out, const float* lse, void* dq, void* dk,
                                 void* dv, int ld_dqkv, float* dscore_sum, int n_chunks, int n_windows, int n_bias_windows,
                                 int heads, float scale, int dtype, void* stream) {
    PSWIN_CHECK_ARG(q && k && v && beta && valid_dtype(dtype) && n_windows > 0 && n_bias_windows > 0 && heads > 0);
    PSWIN_CHECK_ARG(n_windows % n_bias_windows == 0 && ld_qkv >= HD && ld_qkv % 8 == 0 && qkv_head_stride % 8 == 0 && qkv_window_stride % 8 == 0);
    PSWIN_CHECK_ARG(qkv_head_stride >= HD && qkv_window_stride >= (long long)(TOK - 1) * ld_qkv + HD);
    PSWIN_CHECK_ARG(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(dist_t) && aligned16(mask_t));
    PSWIN_CHECK_ARG((long long)n_windows * qkv_window_stride < (1ll << 40));
    PSWIN_CHECK_ARG(!dist_t || (alpha && n_dist > 0 && n_bias_windows % n_dist == 0));
    PSWIN_CHECK_ARG(!mask_t || (n_mask > 0 && n_bias_windows % n_mask == 0));
    PSWIN_CHECK_ARG(dout && lse && dq && dk && dv);
    PSWIN_CHECK_ARG(aligned16(dout) && aligned16(dq) && aligned16(dk) && aligned16(dv) && aligned16(lse));
    PSWIN_CHECK_ARG(ld_out >= heads * HD && ld_out % 8 == 0 && ld_dqkv >= heads * HD && ld_dqkv % 8 == 0);
    const int reps = n_windows / n_bias_windows;
    PSWIN_CHECK_ARG(n_chunks >= 1 && n_chunks <= reps && reps % n_chunks == 0);
    PSWIN_CHECK_ARG(aligned16(dscore_sum));
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v; a.dist = dist_t; a.mask = mask_t; a.alpha = alpha; a.beta = beta;
    a.n_dist = dist_t ? n_dist : 1; a.n_mask = mask_t ? n_mask : 1;
    a.dout = dout; a.lse = const_cast<float*>(lse);
    a.dq = dq; a.dk = dk; a.dv = dv; a.dtab = dscore_sum;
#ifdef PSWIN_ATTN_STAMPS
    a.stamps = g_attn_stamps;
#endif
    a.ld_qkv = ld_qkv; a.ld_out = ld_out; a.ld_dqkv = ld_dqkv;
    a.qkv_win_stride = qkv_window_stride; a.qkv_head_stride = qkv_head_stride;
    a.nb = n_bias_windows; a.heads = heads; a.reps_per_chunk = reps / n_chunks;
    a.n_items = n_chunks * n_bias_windows * heads;
    a.scale = scale;
    if (dtype == PSWIN_BF16) {
        hipLaunchKernelGGL(attn_bwd_pair_kernel, dim3(a.n_items), dim3(128), 0, (hipStream_t)stream, a);
    } else {
        constexpr int W = 3;
        hipLaunchKernelGGL((attn_bwd_kernel<PSWIN_F32, W>), dim3((a.n_items + W - 1) / W), dim3(64 * W), 0,
                           (hipStream_t)stream, a);
    }
    PSWIN_LAUNCH_RET();
}

extern "C" int pswin_attn_bwd(const void* q, const void* k, const void* v, int ld_qkv, const float* dist_t, int n_dist,
                              const float* alpha, const float* beta, const float* mask_t, int n_mask,
                              const void* dout, int ld_out, const float* lse, void* dq, void* dk, void* dv,
                              int ld_dqkv, float* dscore_sum, int n_chunks, int n_windows, int n_bias_windows,
                              int heads, float scale, int dtype, void* stream) {
    PSWIN_CHECK_ARG(ld_qkv >= heads * HD);
    return pswin_attn_bwd_ex(q, k, v, ld_qkv, (long long)TOK * ld_qkv, HD, dist_t, n_dist, alpha, beta, mask_t, n_mask, dout, ld_out, lse,
                             dq, dk, dv, ld_dqkv, dscore_sum, n_chunks, n_windows, n_bias_windows, heads, scale, dtype, stream);
}

extern "C" int pswin_attn_table_grads_batch(const pswin_table_grad_job* jobs, int n_jobs, int stages, void* stream) {
    PSWIN_CHECK_ARG(jobs && n_jobs > 0 && (stages == 1 || stages == 2 || stages == 3 || stages == 4));
    for (int j = 0; j < n_jobs; ++j) {
        const pswin_table_grad_job& q = jobs[j];
        PSWIN_CHECK_ARG(q.dscore_sum && q.dbeta && q.workspace && q.n_tiles > 0 && q.n_bias_windows > 0 && q.heads > 0);
        PSWIN_CHECK_ARG(q.n_tiles % q.n_bias_windows == 0);
        PSWIN_CHECK_ARG(!q.dist_tiles_t || (q.dalpha && q.n_dist > 0 && q.n_bias_windows % q.n_dist == 0));
    }
    for (int at = 0; at < n_jobs; at += TJ_MAX) {
        TBatch b;
        b.n = n_jobs - at < TJ_MAX ? n_jobs - at : TJ_MAX;
        long long g1 = 0, g2 = 0;
        for (int j = 0; j < b.n; ++j) {
            const pswin_table_grad_job& q = jobs[at + j];
            TJob& t = b.job[j];
            // tiles are ordered (chunk, wb, h): tile x of head h belongs to bias window x % n_bias_windows
            t.g = q.dscore_sum; t.dist_t = q.dist_tiles_t; t.partial = q.workspace;
            t.summed = q.workspace + (size_t)DTAB_BLOCKS * dtab_ld(q.heads);
            t.dalpha = q.dist_tiles_t ? q.dalpha : nullptr; t.dbeta = q.dbeta;
            t.n_tiles = q.n_tiles; t.nb = q.n_bias_windows; t.n_dist = q.dist_tiles_t ? q.n_dist : 1; t.heads = q.heads;
            t.blocks = dtab_blocks(q.n_tiles, q.heads);
            t.first1 = (int)g1; t.first2 = (int)g2;
            g1 += (long long)t.blocks * q.heads;
            g2 += (NBINS * q.heads + 3) / 4;
        }
        PSWIN_CHECK_ARG(g1 < 0x7fffffffll && g2 < 0x7fffffffll);
        if (stages & 1)
            hipLaunchKernelGGL(dtab_partial_kernel, dim3((unsigned)g1), dim3(DTAB_THREADS), 0, (hipStream_t)stream, b);
        if (stages & 6) {
            for (int j = 0; (stages & 2) && j < b.n; ++j)
                launch_colsum(b.job[j].partial, b.job[j].blocks, dtab_ld(b.job[j].heads), b.job[j].summed, (hipStream_t)stream);
            hipLaunchKernelGGL(dtab_final_kernel, dim3((unsigned)g2), dim3(256), 0, (hipStream_t)stream, b);
        }
    }
    PSWIN_LAUNCH_RET();
}

extern "C" int pswin_attn_table_grads(const float* dscore_sum, int n_tiles, int n_bias_windows, const float* dist_t,
                                      int n_dist, int heads, float* dalpha, float* dbeta, float* workspace,
                                      void* stream) {
    pswin_table_grad_job q = {dscore_sum, dist_t, dalpha, dbeta, workspace, n_tiles, n_bias_windows, n_dist, heads};
    return pswin_attn_table_grads_batch(&q, 1, 3, stream);
}
