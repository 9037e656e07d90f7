// Streaming GEMM for the Linear layers of the high-resolution stages (gfx950):  Y[M, N] = X[M, K] . W^T (+ bias)
//
// PanoSwin-T's stage-0 projections (qkv / proj / fc1 / fc2, HOT:287, 309, 50-58) have K, N <= 384 against M = 262k-276k
// rows: they are HBM-bound streaming operators (212 MB moved for 15 GFLOP in qkv), yet the library's general GEMM
// kernels reach only 1.9-3 TB/s on them (qkv forward: 113 us in the step, 45 us of traffic at 4.7 TB/s).  Here the
// whole weight matrix (<= 80 KB) is staged ONCE per workgroup into LDS as MFMA A operands (rows = output column,
// contraction contiguous; 16 bytes of row padding make the 16-byte reads of 8 rows conflict-free), every wave streams
// 32 (16) rows of X straight from HBM in MFMA B-operand layout (a row's 16-byte chunk IS the operand fragment: no
// staging, no transposes; rows beyond M return zeros through the buffer range check), and writes its bf16 output
// rows with 16-byte stores (accumulator quads of two column tiles exchanged across lane groups).  The same kernel
// computes the data gradient dX = dY . W by staging the transposed weight (transpose done while writing the LDS image).
// bf16 operands, f32 accumulation on v_mfma_f32_16x16x32_bf16: the arithmetic of the library path.
#include "pswin_common.hpp"
#include "pswin_gelu.hpp"

using namespace pswin;

namespace {

constexpr int THREADS = 256;
using rsrc_t = __amdgpu_buffer_rsrc_t;

__device__ inline f32x4 mfma32(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ inline void swap16_u32(unsigned& a, unsigned& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
__device__ inline unsigned pack_bf16(float lo, float hi) {
    return pack2_bf16(lo, hi);
}
// lane (c, g) holds quads q0 = col[4g..4g+3], q1 = col[16+4g..16+4g+3] of a 32-column group of one row -> after the
// exchange 8 consecutive columns starting at 8 (g >> 1) + 16 (g & 1)
__device__ inline u32x4 pack_row8(f32x4 q0, f32x4 q1) {
    unsigned a0 = pack_bf16(q0[0], q0[1]), a1 = pack_bf16(q0[2], q0[3]);
    unsigned b0 = pack_bf16(q1[0], q1[1]), b1 = pack_bf16(q1[2], q1[3]);
    swap16_u32(a0, b0);
    swap16_u32(a1, b1);
    return u32x4{a0, a1, b0, b1};
}

// the same exchange on f32 quads (whole-vector bit casts: hipcc folds per-element casts of vector lanes)
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

// Epilogues.  EPI 0: y = acc + bias.  EPI 1: y = gelu(acc + bias) -- fc1 + bias + nn.GELU (HOT:50-57) in one pass, the
// pre-activation is never stored.  EPI 2 (backward of EPI 1): the pre-activation is RECOMPUTED (K = 96: cheaper than
// storing and re-reading 201 MB), y = aux * gelu'(acc + bias) with aux = dL/d gelu-output, and the per-column sums of
// y (the fc1 bias gradient) are accumulated per workgroup into `partial` [gridDim.x][N] (fixed order).
// KS = K / 32 contraction steps, NT = N / 16 output-column tiles (even), RT = 16-row tiles per wave iteration
template <int KS, int NT, int RT, int EPI>
__global__ __launch_bounds__(THREADS, 2) void skinny_gemm_kernel(const void* __restrict__ x, const void* __restrict__ w,
                                                                 const float* __restrict__ bias, void* __restrict__ y,
                                                                 int M, int transpose_w, const void* __restrict__ aux,
                                                                 float* __restrict__ partial) {
    constexpr int K = 32 * KS, N = 16 * NT;
    // LDS row stride in bytes.  K = 96: dense 192-byte rows with the 16-byte chunk XOR-swizzled inside its group of 4 by
    // row bits 1-2 (keeps the largest image, 384 x 96, at 72 KB so that two workgroups fit a CU); otherwise 16 bytes
    // of padding.  Both make the 16-byte reads of 8 consecutive rows conflict-free.
    constexpr int LD = (K == 96) ? 192 : 2 * K + 16;
    auto woff = [](int row, int chunk) {
        if constexpr (K == 96) return row * 192 + ((((chunk & ~3) | ((chunk ^ (row >> 1)) & 3))) << 4);
        else return row * LD + chunk * 16;
    };
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wl = smem;                                                     // [N][LD]
    float* bl = reinterpret_cast<float*>(smem + N * LD);                 // [N]
    float* cs = bl + N;                                                  // EPI 2: [4 waves][N] column sums
    if (!transpose_w) {                                                  // w: [N][K] (nn.Linear layout)
        for (int i = threadIdx.x; i < N * (K / 8); i += THREADS) {
            const int row = i / (K / 8), ch = i - row * (K / 8);
            *reinterpret_cast<u32x4*>(wl + woff(row, ch)) = reinterpret_cast<const u32x4*>(w)[i];
        }
    } else {                                                             // w: [K][N] -> image[n][k]
        const unsigned short* ws = reinterpret_cast<const unsigned short*>(w);
        for (int i = threadIdx.x; i < K * (N / 8); i += THREADS) {
            const int k = i / (N / 8), n0 = (i - k * (N / 8)) * 8;
            const u32x4 v = *reinterpret_cast<const u32x4*>(ws + (size_t)k * N + n0);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                *reinterpret_cast<unsigned short*>(wl + woff(n0 + 2 * d, k >> 3) + 2 * (k & 7)) = (unsigned short)(v[d] & 0xffffu);
                *reinterpret_cast<unsigned short*>(wl + woff(n0 + 2 * d + 1, k >> 3) + 2 * (k & 7)) = (unsigned short)(v[d] >> 16);
            }
        }
    }
    for (int i = threadIdx.x; i < N; i += THREADS) bl[i] = bias ? bias[i] : 0.f;
    if constexpr (EPI == 2)
        for (int i = threadIdx.x; i < 4 * N; i += THREADS) cs[i] = 0.f;
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
    const rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(x), 0, (int)((size_t)M * K * 2), 0x00020000);
    const rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(y, 0, (int)((size_t)M * N * 2), 0x00020000);
    const rsrc_t as = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(aux), 0, EPI == 2 ? (int)((size_t)M * N * 2) : 0, 0x00020000);
    constexpr int ROWS = 16 * RT;
    const int ntiles = (M + ROWS - 1) / ROWS;
    const int d0 = 8 * (g >> 1) + 16 * (g & 1);
    auto load_b = [&](int tile, bf16x8 (&b)[RT][KS]) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const unsigned row = (unsigned)tile * ROWS + 16 * rt + c;
            const unsigned off = row < (unsigned)M ? row * (unsigned)(K * 2) + 16u * g : 0xFFFFFF00u;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const unsigned o = off == 0xFFFFFF00u ? off : off + 64u * s;
                b[rt][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xs, o, 0, 0));
            }
        }
    };
    // The activation rows of a wave's NEXT tile are requested as soon as the MFMAs of the current one have consumed the
    // registers, i.e. before the epilogue's stores: loads issued behind stores wait for them (one in-order counter), which
    // left each wave with nothing in flight between its last store and the arrival of its next rows.
    const int tile0 = blockIdx.x * (THREADS / 64) + wave, tstep = gridDim.x * (THREADS / 64);
    bf16x8 b[RT][KS];
    auto row_tile = [&](int tile) {
        f32x4 acc[RT][NT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[rt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // weight fragments in batches of 6 column tiles, double buffered in registers: the next batch's LDS reads are in
        // flight under the current batch's MFMAs; the scheduling barriers keep the compiler from hoisting all KS * NT
        // reads to the top (which spilled ~290 registers)
        constexpr int BT = 6, NB = KS * (NT / BT);
        static_assert(NT % BT == 0, "column tiles come in batches of 6");
        bf16x8 a[2][BT];
        auto read_batch = [&](int bi, bf16x8 (&dst)[BT]) {
            const int s_ = bi / (NT / BT), nt0 = (bi - s_ * (NT / BT)) * BT;
#pragma unroll
            for (int j = 0; j < BT; ++j)
                dst[j] = *reinterpret_cast<const bf16x8*>(wl + woff(16 * (nt0 + j) + c, 4 * s_ + g));
        };
        read_batch(0, a[0]);
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) {
            if (bi + 1 < NB) read_batch(bi + 1, a[(bi + 1) & 1]);
            const int s_ = bi / (NT / BT), nt0 = (bi - s_ * (NT / BT)) * BT;
#pragma unroll
            for (int j = 0; j < BT; ++j)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt][nt0 + j] = mfma32(a[bi & 1][j], b[rt][s_], acc[rt][nt0 + j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // EPI 2: every dL/dh load of the tile before its first store, for the same reason
        [[maybe_unused]] u32x4 dhv[RT][NT / 2];
        if constexpr (EPI == 2) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const unsigned row = (unsigned)tile * ROWS + 16 * rt + c;
                const unsigned base = row < (unsigned)M ? row * (unsigned)(N * 2) + 2u * d0 : 0xFFFFFF00u;
#pragma unroll
                for (int np = 0; np < NT / 2; ++np)
                    dhv[rt][np] = __builtin_amdgcn_raw_buffer_load_b128(as, base == 0xFFFFFF00u ? base : base + 64u * np, 0, 0);   // rows >= M: zeros
            }
        }
        if (tile + tstep < ntiles) load_b(tile + tstep, b);
        __builtin_amdgcn_sched_barrier(0);
        // acc[rt][nt][e] = Y[row 16 rt + c][column 16 nt + 4 g + e]
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const unsigned row = (unsigned)tile * ROWS + 16 * rt + c;
            const unsigned base = row < (unsigned)M ? row * (unsigned)(N * 2) + 2u * d0 : 0xFFFFFF00u;
#pragma unroll
            for (int np = 0; np < NT / 2; ++np) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(bl + 32 * np + 4 * g);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(bl + 32 * np + 16 + 4 * g);
                f32x4 q0 = acc[rt][2 * np] + b0, q1 = acc[rt][2 * np + 1] + b1;
                const unsigned off = base == 0xFFFFFF00u ? base : base + 64u * np;
                if constexpr (EPI == 1) {
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const gelu_f32x2 a = gelu_f2(gelu_f32x2{q0[e], q0[e + 1]}), b = gelu_f2(gelu_f32x2{q1[e], q1[e + 1]});
                        q0[e] = a[0]; q0[e + 1] = a[1];
                        q1[e] = b[0]; q1[e + 1] = b[1];
                    }
                }
                if constexpr (EPI == 2) {
                    // 8 consecutive columns 32 np + d0 .. of this lane's row, the layout of the 16-byte aux load
                    exchange_row8(q0, q1);
                    const u32x4 dh = dhv[rt][np];
                    float v[8];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const gelu_f32x2 yy = d < 2 ? gelu_f32x2{q0[2 * d], q0[2 * d + 1]} : gelu_f32x2{q1[2 * d - 4], q1[2 * d - 3]};
                        const gelu_f32x2 gg = gelu_grad_f2(yy);
                        v[2 * d] = __builtin_bit_cast(float, dh[d] << 16) * gg[0];
                        v[2 * d + 1] = __builtin_bit_cast(float, dh[d] & 0xffff0000u) * gg[1];
                    }
                    const u32x4 o = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
                    __builtin_amdgcn_raw_buffer_store_b128(o, ys, off, 0, 0);
                    // column sums over the 16 rows of the tile (lanes c of a group share the columns), then one lane per
                    // group adds them to this wave's LDS row: the kernel is HBM bound, the VALU has the slack
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float sum = row16_sum(v[j]);
                        if (c == 0) cs[wave * N + 32 * np + d0 + j] += sum;
                    }
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(pack_row8(q0, q1), ys, off, 0, 0);
                }
            }
        }
    };
    // The first tile is peeled off the loop: the loop header is then reached only with "next rows requested, then this tile's
    // stores" outstanding on both edges, and the wait in front of the first MFMA counts past the stores (with the first load
    // in the preheader the merged state made it wait for every store).
    if (tile0 < ntiles) {
        load_b(tile0, b);
        row_tile(tile0);
        for (int tile = tile0 + tstep; tile < ntiles; tile += tstep) row_tile(tile);
    }
    if constexpr (EPI == 2) {
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += THREADS)
            partial[(size_t)blockIdx.x * N + i] = (cs[i] + cs[N + i]) + (cs[2 * N + i] + cs[3 * N + i]);
    }
}

constexpr int MAX_GRID = 512;                    // 2 workgroups per CU, persistent over the row tiles

template <int KS, int NT, int RT, int EPI>
int launch(const void* x, const void* w, const float* bias, void* y, int M, int transpose_w, const void* aux, float* partial,
           hipStream_t st, int* grid_out = nullptr) {
    constexpr int K = 32 * KS, N = 16 * NT;
    constexpr size_t lds = (size_t)N * (K == 96 ? 192 : 2 * K + 16) + N * sizeof(float) * (EPI == 2 ? 5 : 1);
    static std::atomic<unsigned long long> configured{0};
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&skinny_gemm_kernel<KS, NT, RT, EPI>), lds, configured)) return rc;
    const int ntiles = (M + 16 * RT - 1) / (16 * RT);
    int grid = (ntiles + 3) / 4;
    if (grid > MAX_GRID) grid = MAX_GRID;
    if (grid_out) *grid_out = grid;
    hipLaunchKernelGGL((skinny_gemm_kernel<KS, NT, RT, EPI>), dim3(grid), dim3(THREADS), lds, st, x, w, bias, y, M, transpose_w,
                       aux, partial);
    PSWIN_LAUNCH_RET();
}


// ---- stage-0 Mlp backward, first half, in one pass (round 3) ----------------------------------------------------------------------
// g = (dy . W2) * gelu'(x . W1^T + b1)   -- fc2's data gradient (HOT:58) and the backward of nn.GELU (HOT:57) with fc1's
// pre-activation recomputed (HOT:56), + the per-workgroup column sums of g (the fc1 bias gradient).  As two launches this was the
// library's slowest remaining GEMM (83 us: [262144, 96] x [96, 384]) writing dh = dy . W2 (201 MB at batch 8) and pswin_fc1_gelu_bwd
// reading it back; here dh exists only in accumulators.  Both weights are resident in LDS as MFMA A-operand row fragments (W1 as it
// is, W2 transposed while it is staged: 2 x 72 KB, one 8-wave workgroup per CU), a wave streams 16 rows of x and of dy (a row's
// 16-byte chunk is a B fragment) and runs the 384 output columns in two halves of 12 column tiles, two accumulator sets (pre, dh)
// of 48 registers each; weight fragments in batches of 6, double buffered in registers as in skinny_gemm_kernel.
constexpr int M0_THREADS = 512, M0_WAVES = 8, M0_K = 96, M0_N = 384, M0_KS = 3, M0_HT = 12;
constexpr int M0_LDS = 2 * M0_N * 192 + M0_N * 4 + M0_WAVES * M0_N * 4;

template <bool SUMS>
__global__ __launch_bounds__(M0_THREADS, 2) void mlp0_bwd_kernel(const void* __restrict__ x, const void* __restrict__ w1, const float* __restrict__ b1,
                                                                 const void* __restrict__ dy, const void* __restrict__ w2, void* __restrict__ gout,
                                                                 float* __restrict__ partial, int M) {
    constexpr int K = M0_K, N = M0_N, KS = M0_KS;
    auto woff = [](int row, int chunk) { return row * 192 + ((((chunk & ~3) | ((chunk ^ (row >> 1)) & 3))) << 4); };
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* w1l = smem;                                                    // [N][192]: W1 rows (output column n, contraction k)
    char* w2l = smem + N * 192;                                          // [N][192]: W2^T rows (hidden column n, contraction k2)
    float* bl = reinterpret_cast<float*>(smem + 2 * N * 192);            // [N]
    float* cs = bl + N;                                                  // [8 waves][N] column sums
    for (int i = threadIdx.x; i < N * (K / 8); i += M0_THREADS) {
        const int row = i / (K / 8), ch = i - row * (K / 8);
        *reinterpret_cast<u32x4*>(w1l + woff(row, ch)) = reinterpret_cast<const u32x4*>(w1)[i];
    }
    {
        const unsigned short* ws = reinterpret_cast<const unsigned short*>(w2);      // fc2.weight [96][384] -> image[n][k2]
        for (int i = threadIdx.x; i < K * (N / 8); i += M0_THREADS) {
            const int k = i / (N / 8), n0 = (i - k * (N / 8)) * 8;
            const u32x4 v = *reinterpret_cast<const u32x4*>(ws + (size_t)k * N + n0);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                *reinterpret_cast<unsigned short*>(w2l + woff(n0 + 2 * d, k >> 3) + 2 * (k & 7)) = (unsigned short)(v[d] & 0xffffu);
                *reinterpret_cast<unsigned short*>(w2l + woff(n0 + 2 * d + 1, k >> 3) + 2 * (k & 7)) = (unsigned short)(v[d] >> 16);
            }
        }
    }
    for (int i = threadIdx.x; i < N; i += M0_THREADS) bl[i] = b1 ? b1[i] : 0.f;
    if constexpr (SUMS)
        for (int i = threadIdx.x; i < M0_WAVES * N; i += M0_THREADS) cs[i] = 0.f;
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
    const rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(x), 0, (int)((size_t)M * K * 2), 0x00020000);
    const rsrc_t ds = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(dy), 0, (int)((size_t)M * K * 2), 0x00020000);
    const rsrc_t gs = __builtin_amdgcn_make_buffer_rsrc(gout, 0, (int)((size_t)M * N * 2), 0x00020000);
    const int ntiles = (M + 15) / 16;
    const int d0 = 8 * (g >> 1) + 16 * (g & 1);
    const int lane_off = c * 192 + (((g ^ (c >> 1)) & 3) << 4);
    bf16x8 bx[KS], bd[KS];
    auto load_b = [&](int tile) {
        const unsigned row = (unsigned)tile * 16 + c;
        const unsigned off = row < (unsigned)M ? row * (unsigned)(K * 2) + 16u * g : 0xFFFFFF00u;          // rows >= M: zeros
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const unsigned o = off == 0xFFFFFF00u ? off : off + 64u * s;
            bx[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xs, o, 0, 0));
            bd[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ds, o, 0, 0));
        }
    };
    const int tile0 = blockIdx.x * M0_WAVES + wave, tstep = gridDim.x * M0_WAVES;
    auto row_tile = [&](int tile) {
        const unsigned row = (unsigned)tile * 16 + c;
        const unsigned base = row < (unsigned)M ? row * (unsigned)(N * 2) + 2u * d0 : 0xFFFFFF00u;
        // the bias quads are the same for every tile: left to itself the compiler reads all 96 values once in front of the tile
        // loop and spills them; an opaque zero in the address keeps the (cheap) LDS reads in the epilogue
        int zb = 0;
        asm volatile("" : "+v"(zb));
        const float* blt = bl + zb;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 ap[M0_HT], ad[M0_HT];
#pragma unroll
            for (int t = 0; t < M0_HT; ++t) {
                ap[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                ad[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            // per contraction step the half's 24 fragments [W1 tiles 0..11 | W2^T tiles 0..11] in batches of 6
            constexpr int BT = 6, BPS = 2 * M0_HT / BT, NB = KS * BPS;
            bf16x8 a[2][BT];
            // fragment (column tile t of this half, step s) of an image: lane part + 3072 t + 64 s -- woff(16 nt + c, 4 s + g) written out,
            // with one base per (image, half) so that every immediate stays below the 64 KB of a ds_read offset (bases the compiler
            // derives itself for the second image were spilled and reloaded per tile, each reload waiting out the previous tile's stores)
            const char* img1 = w1l + lane_off + half * (M0_HT * 3072);
            const char* img2 = w2l + lane_off + half * (M0_HT * 3072);
            auto read_batch = [&](int bi, bf16x8 (&dst)[BT]) {
                const int s_ = bi / BPS, e0 = (bi - s_ * BPS) * BT;
#pragma unroll
                for (int j = 0; j < BT; ++j) {
                    const int e = e0 + j;
                    dst[j] = *reinterpret_cast<const bf16x8*>((e < M0_HT ? img1 : img2) + 3072 * (e < M0_HT ? e : e - M0_HT) + 64 * s_);
                }
            };
            read_batch(0, a[0]);
#pragma unroll
            for (int bi = 0; bi < NB; ++bi) {
                if (bi + 1 < NB) read_batch(bi + 1, a[(bi + 1) & 1]);
                const int s_ = bi / BPS, e0 = (bi - s_ * BPS) * BT;
#pragma unroll
                for (int j = 0; j < BT; ++j) {
                    const int e = e0 + j;
                    if (e < M0_HT) ap[e] = mfma32(a[bi & 1][j], bx[s_], ap[e]);
                    else ad[e - M0_HT] = mfma32(a[bi & 1][j], bd[s_], ad[e - M0_HT]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // the next tile's rows are requested once the last MFMA has consumed the registers, before this half's stores
            if (half == 1 && tile + tstep < ntiles) load_b(tile + tstep);
            __builtin_amdgcn_sched_barrier(0);
            // ap[t][e] = pre-activation - b1, ad[t][e] = dh at [row c][column 192 half + 16 t + 4 g + e]
#pragma unroll
            for (int np = 0; np < M0_HT / 2; ++np) {
                const int col = 192 * half + 32 * np;
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(blt + col + 4 * g);
                const f32x4 b1v = *reinterpret_cast<const f32x4*>(blt + col + 16 + 4 * g);
                const f32x4 p0 = ap[2 * np] + b0, p1 = ap[2 * np + 1] + b1v;
                f32x4 v0, v1;
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    // the forward pass rounds the pre-activation to nothing (it is never stored) and h to bf16; dh is an f32 accumulator
                    const gelu_f32x2 g0 = gelu_grad_f2(gelu_f32x2{p0[e], p0[e + 1]}), g1 = gelu_grad_f2(gelu_f32x2{p1[e], p1[e + 1]});
                    v0[e] = ad[2 * np][e] * g0[0];
                    v0[e + 1] = ad[2 * np][e + 1] * g0[1];
                    v1[e] = ad[2 * np + 1][e] * g1[0];
                    v1[e + 1] = ad[2 * np + 1][e + 1] * g1[1];
                }
                const unsigned off = base == 0xFFFFFF00u ? base : base + 2u * col;
                __builtin_amdgcn_raw_buffer_store_b128(pack_row8(v0, v1), gs, off, 0, 0);
                // column sums over the tile's 16 rows (the lanes c of a group share the columns); rows >= M contribute zeros (their
                // dy rows read as zeros, so dh = 0).  32 DPP adds per column pair: the caller that runs the fc1 weight gradient on
                // pswin_gemm_tn_ring_bias gets the same sums from its MFMAs and asks for none here (workspace = NULL).
                if constexpr (SUMS) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float s0 = row16_sum(v0[e]), s1 = row16_sum(v1[e]);
                        if (c == 0) {
                            cs[wave * N + col + 4 * g + e] += s0;
                            cs[wave * N + col + 16 + 4 * g + e] += s1;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);               // one column pair at a time (interleaved, the six pairs' temporaries spill)
            }
        }
    };
    if (tile0 < ntiles) {
        load_b(tile0);
        row_tile(tile0);
        for (int tile = tile0 + tstep; tile < ntiles; tile += tstep) row_tile(tile);
    }
    if constexpr (SUMS) {
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += M0_THREADS) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < M0_WAVES; ++w) sum += cs[w * N + i];
            partial[(size_t)blockIdx.x * N + i] = sum;
        }
    }
}

// ---- stage-0 Mlp forward in one pass (round 3):  h = gelu(x W1^T + b1) (stored: the backward pass reads it), y = h W2^T ----------------
// The second product takes h straight from the first one's accumulators: packing the two 16-column accumulator tiles of a 32-column
// block gives the 32-deep B fragment in a permuted but fixed column order (slot 8 g + j = column 4 g + j for j < 4, 16 + 4 g + j - 4
// otherwise -- the order pack8 / pswin_fused.hip use), and W2 is staged into LDS in that order, so h is written once and not read back
// (201 MB per block at batch 8) and the fc2 launch is gone.  LDS: W1 image [384][192 B] + W2 image [96][768 B, chunk ^ (row & 15)].
__global__ __launch_bounds__(M0_THREADS, 2) void mlp0_fwd_kernel(const void* __restrict__ x, const void* __restrict__ w1, const float* __restrict__ b1,
                                                                 const void* __restrict__ w2, void* __restrict__ hout, void* __restrict__ yout, int M) {
    constexpr int K = M0_K, N = M0_N, KS = M0_KS, OT = M0_K / 16;
    auto woff = [](int row, int chunk) { return row * 192 + ((((chunk & ~3) | ((chunk ^ (row >> 1)) & 3))) << 4); };
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* w1l = smem;                                                    // [N][192]: W1 rows (hidden column n, contraction k)
    char* w2l = smem + N * 192;                                          // [96][768]: W2 rows (output column o, contraction = hidden, permuted)
    float* bl = reinterpret_cast<float*>(smem + 2 * N * 192);            // [N]
    for (int i = threadIdx.x; i < N * (K / 8); i += M0_THREADS) {
        const int row = i / (K / 8), ch = i - row * (K / 8);
        *reinterpret_cast<u32x4*>(w1l + woff(row, ch)) = reinterpret_cast<const u32x4*>(w1)[i];
    }
    {
        // fc2.weight [96][384]: 8-byte piece (o, block np, quad q = 0..7 of 4 columns) goes to chunk 4 np + (q & 3), half q >> 2
        const unsigned long long* ws = reinterpret_cast<const unsigned long long*>(w2);
        for (int i = threadIdx.x; i < K * (N / 4); i += M0_THREADS) {
            const int o = i / (N / 4), q4 = i - o * (N / 4), np = q4 >> 3, qq = q4 & 7;
            const int chunk = 4 * np + (qq & 3);
            *reinterpret_cast<unsigned long long*>(w2l + o * 768 + ((chunk ^ (o & 15)) << 4) + 8 * (qq >> 2)) = ws[i];
        }
    }
    for (int i = threadIdx.x; i < N; i += M0_THREADS) bl[i] = b1 ? b1[i] : 0.f;
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
    const rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(x), 0, (int)((size_t)M * K * 2), 0x00020000);
    const rsrc_t hs = __builtin_amdgcn_make_buffer_rsrc(hout, 0, (int)((size_t)M * N * 2), 0x00020000);
    const rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(yout, 0, (int)((size_t)M * K * 2), 0x00020000);
    const int ntiles = (M + 15) / 16;
    const int d0 = 8 * (g >> 1) + 16 * (g & 1);
    const int lane_off = c * 192 + (((g ^ (c >> 1)) & 3) << 4);
    // W2 fragment (output tile ot, block np): row 16 ot + c, chunk (4 np + g) ^ c = 16 (np >> 2) + ((4 (np & 3) + g) ^ c)
    const char* w2b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w2b[j] = w2l + c * 768 + (((4 * j + g) ^ c) << 4);
    bf16x8 bx[KS];
    auto load_b = [&](int tile) {
        const unsigned row = (unsigned)tile * 16 + c;
        const unsigned off = row < (unsigned)M ? row * (unsigned)(K * 2) + 16u * g : 0xFFFFFF00u;          // rows >= M: zeros
#pragma unroll
        for (int s = 0; s < KS; ++s) bx[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xs, off == 0xFFFFFF00u ? off : off + 64u * s, 0, 0));
    };
    const int tile0 = blockIdx.x * M0_WAVES + wave, tstep = gridDim.x * M0_WAVES;
    auto row_tile = [&](int tile) {
        const unsigned row = (unsigned)tile * 16 + c;
        const unsigned hbase = row < (unsigned)M ? row * (unsigned)(N * 2) + 2u * d0 : 0xFFFFFF00u;
        const unsigned ybase = row < (unsigned)M ? row * (unsigned)(K * 2) + 2u * d0 : 0xFFFFFF00u;
        int zb = 0;                                                      // (keeps the bias reads in the epilogue: see mlp0_bwd_kernel)
        asm volatile("" : "+v"(zb));
        const float* blt = bl + zb;
        f32x4 ay[OT];
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) ay[ot] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 ap[M0_HT];
#pragma unroll
            for (int t = 0; t < M0_HT; ++t) ap[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            const char* img1 = w1l + lane_off + half * (M0_HT * 3072);
            constexpr int BT = 6, BPS = M0_HT / BT, NB = KS * BPS;
            bf16x8 a[2][BT];
            auto read_batch = [&](int bi, bf16x8 (&dst)[BT]) {
                const int s_ = bi / BPS, e0 = (bi - s_ * BPS) * BT;
#pragma unroll
                for (int j = 0; j < BT; ++j) dst[j] = *reinterpret_cast<const bf16x8*>(img1 + 3072 * (e0 + j) + 64 * s_);
            };
            read_batch(0, a[0]);
#pragma unroll
            for (int bi = 0; bi < NB; ++bi) {
                if (bi + 1 < NB) read_batch(bi + 1, a[(bi + 1) & 1]);
                const int s_ = bi / BPS, e0 = (bi - s_ * BPS) * BT;
#pragma unroll
                for (int j = 0; j < BT; ++j) ap[e0 + j] = mfma32(a[bi & 1][j], bx[s_], ap[e0 + j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (half == 1 && tile + tstep < ntiles) load_b(tile + tstep);       // next tile's rows before this half's stores
            __builtin_amdgcn_sched_barrier(0);
            // one 32-column block at a time: bias + GELU, the bf16 fragment, its 6 products into y, the h store
#pragma unroll
            for (int np = 0; np < M0_HT / 2; ++np) {
                const int blk = 6 * half + np, col = 32 * blk;
                bf16x8 w2f[OT];
#pragma unroll
                for (int ot = 0; ot < OT; ++ot) w2f[ot] = *reinterpret_cast<const bf16x8*>(w2b[blk & 3] + 16 * ot * 768 + (blk >> 2) * 256);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(blt + col + 4 * g);
                const f32x4 b1v = *reinterpret_cast<const f32x4*>(blt + col + 16 + 4 * g);
                f32x4 q0 = ap[2 * np] + b0, q1 = ap[2 * np + 1] + b1v;
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const gelu_f32x2 h0 = gelu_f2(gelu_f32x2{q0[e], q0[e + 1]}), h1 = gelu_f2(gelu_f32x2{q1[e], q1[e + 1]});
                    q0[e] = h0[0]; q0[e + 1] = h0[1];
                    q1[e] = h1[0]; q1[e + 1] = h1[1];
                }
                unsigned a0 = pack_bf16(q0[0], q0[1]), a1 = pack_bf16(q0[2], q0[3]), c0 = pack_bf16(q1[0], q1[1]), c1 = pack_bf16(q1[2], q1[3]);
                const bf16x8 hf = __builtin_bit_cast(bf16x8, u32x4{a0, a1, c0, c1});
#pragma unroll
                for (int ot = 0; ot < OT; ++ot) ay[ot] = mfma32(w2f[ot], hf, ay[ot]);
                swap16_u32(a0, c0);
                swap16_u32(a1, c1);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{a0, a1, c0, c1}, hs, hbase == 0xFFFFFF00u ? hbase : hbase + 2u * col, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ay[ot][e] = y[row c][column 16 ot + 4 g + e]
#pragma unroll
        for (int p2 = 0; p2 < OT / 2; ++p2)
            __builtin_amdgcn_raw_buffer_store_b128(pack_row8(ay[2 * p2], ay[2 * p2 + 1]), ys, ybase == 0xFFFFFF00u ? ybase : ybase + 64u * p2, 0, 0);
    };
    if (tile0 < ntiles) {
        load_b(tile0);
        row_tile(tile0);
        for (int tile = tile0 + tstep; tile < ntiles; tile += tstep) row_tile(tile);
    }
}

constexpr int M0_MAX_GRID = 256;                 // one 8-wave workgroup per CU (147 KB of weights in LDS)
inline int mlp0_grid(long long M) {
    const long long ntiles = (M + 15) / 16, g = (ntiles + M0_WAVES - 1) / M0_WAVES;
    return (int)(g > M0_MAX_GRID ? M0_MAX_GRID : g);
}

}  // namespace

extern "C" {

int pswin_gemm_skinny_supported(int K, int N) {
    return (K == 96 && (N == 96 || N == 288 || N == 384)) || (K == 288 && N == 96) || (K == 384 && N == 96) ||
           (K == 192 && N == 192);
}

int pswin_gemm_skinny(const void* x, const void* w, const float* bias, void* y, long long M, int K, int N, int transpose_w,
                      void* stream) {
    PSWIN_CHECK_ARG(x && w && y && M > 0 && pswin_gemm_skinny_supported(K, N));
    PSWIN_CHECK_ARG(M * (long long)(K > N ? K : N) * 2 < 0xFFFFFF00ll && aligned16(x) && aligned16(w) && aligned16(y));
    hipStream_t st = (hipStream_t)stream;
    const int m = (int)M;
    if (K == 96 && N == 288) return launch<3, 18, 2, 0>(x, w, bias, y, m, transpose_w, nullptr, nullptr, st);
    if (K == 96 && N == 96) return launch<3, 6, 2, 0>(x, w, bias, y, m, transpose_w, nullptr, nullptr, st);
    if (K == 96 && N == 384) return launch<3, 24, 1, 0>(x, w, bias, y, m, transpose_w, nullptr, nullptr, st);
    if (K == 288 && N == 96) return launch<9, 6, 2, 0>(x, w, bias, y, m, transpose_w, nullptr, nullptr, st);
    if (K == 384 && N == 96) return launch<12, 6, 2, 0>(x, w, bias, y, m, transpose_w, nullptr, nullptr, st);
    if (K == 192 && N == 192) return launch<6, 12, 2, 0>(x, w, bias, y, m, transpose_w, nullptr, nullptr, st);
    return PSWIN_ERR_ARG;
}

/* fc1 + bias + GELU of the stage-0 Mlp (K = 96, N = 384) */
int pswin_fc1_gelu_supported(int K, int N) { return K == 96 && N == 384; }

int pswin_fc1_gelu_fwd(const void* x, const void* w, const float* bias, void* h, long long M, int K, int N, void* stream) {
    PSWIN_CHECK_ARG(x && w && h && M > 0 && pswin_fc1_gelu_supported(K, N));
    PSWIN_CHECK_ARG(M * (long long)N * 2 < 0xFFFFFF00ll && aligned16(x) && aligned16(w) && aligned16(h));
    return launch<3, 24, 1, 1>(x, w, bias, h, (int)M, 0, nullptr, nullptr, (hipStream_t)stream);
}

int pswin_fc1_gelu_workspace(int N) { return N > 0 ? MAX_GRID * N : PSWIN_ERR_ARG; }

int pswin_fc1_gelu_bwd(const void* x, const void* w, const float* bias, const void* dh, void* dy, float* dbias,
                       float* workspace, long long M, int K, int N, void* stream) {
    PSWIN_CHECK_ARG(x && w && dh && dy && workspace && M > 0 && pswin_fc1_gelu_supported(K, N));
    PSWIN_CHECK_ARG(M * (long long)N * 2 < 0xFFFFFF00ll && aligned16(x) && aligned16(w) && aligned16(dh) && aligned16(dy));
    int grid = 0;
    const int rc = launch<3, 24, 1, 2>(x, w, bias, dy, (int)M, 0, dh, workspace, (hipStream_t)stream, &grid);
    if (rc) return rc;
    if (dbias) launch_colsum(workspace, grid, N, dbias, (hipStream_t)stream);   // else: partial rows only
    PSWIN_LAUNCH_RET();
}

/* Stage-0 Mlp backward, first half (autograd of HOT:56-58): g = (dy . W2) * gelu'(x . W1^T + b1) in one pass, dh never stored.
 * x [M, C], dy [M, C] bf16 rows; w1 = fc1.weight [hidden, C], w2 = fc2.weight [C, hidden] bf16; b1 f32 [hidden]; g [M, hidden] bf16;
 * workspace f32 [pswin_mlp0_bwd_partial_rows(M)][hidden] receives the per-workgroup column sums of g, dbias1 (or NULL: partial rows
 * only) their fixed-order sum; workspace = NULL: no column sums at all (the caller takes them from pswin_gemm_tn_ring_bias). */
int pswin_mlp0_bwd_supported(int C, int hidden) { return C == M0_K && hidden == M0_N; }

int pswin_mlp0_bwd_partial_rows(long long M) {
    if (M <= 0 || M > 0x7fffffffll) return PSWIN_ERR_ARG;
    return mlp0_grid(M);
}

int pswin_mlp0_bwd(const void* x, const void* w1, const float* b1, const void* dy, const void* w2, void* g, float* dbias1, float* workspace,
                   long long M, int C, int hidden, void* stream) {
    PSWIN_CHECK_ARG(x && w1 && dy && w2 && g && M > 0 && pswin_mlp0_bwd_supported(C, hidden) && (workspace || !dbias1));
    PSWIN_CHECK_ARG(M * (long long)hidden * 2 < 0xFFFFFF00ll && aligned16(x) && aligned16(w1) && aligned16(dy) && aligned16(w2) && aligned16(g));
    const int grid = mlp0_grid(M);
    const hipStream_t st = (hipStream_t)stream;
    if (workspace) {
        static std::atomic<unsigned long long> configured{0};
        if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&mlp0_bwd_kernel<true>), M0_LDS, configured)) return rc;
        hipLaunchKernelGGL(mlp0_bwd_kernel<true>, dim3(grid), dim3(M0_THREADS), M0_LDS, st, x, w1, b1, dy, w2, g, workspace, (int)M);
        if (dbias1) launch_colsum(workspace, grid, hidden, dbias1, st);
    } else {
        static std::atomic<unsigned long long> configured{0};
        if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&mlp0_bwd_kernel<false>), M0_LDS, configured)) return rc;
        hipLaunchKernelGGL(mlp0_bwd_kernel<false>, dim3(grid), dim3(M0_THREADS), M0_LDS, st, x, w1, b1, dy, w2, g, workspace, (int)M);
    }
    PSWIN_LAUNCH_RET();
}

/* Stage-0 Mlp forward in one pass (HOT:50-58 without fc2's bias): h[M, hidden] = gelu(x W1^T + b1) (kept for the backward pass),
 * y[M, C] = h W2^T from the accumulators of the first product.  x [M, C] bf16; w1 [hidden, C], w2 [C, hidden] bf16; b1 f32 or NULL. */
int pswin_mlp0_fwd(const void* x, const void* w1, const float* b1, const void* w2, void* h, void* y, long long M, int C, int hidden, void* stream) {
    PSWIN_CHECK_ARG(x && w1 && w2 && h && y && M > 0 && pswin_mlp0_bwd_supported(C, hidden));
    PSWIN_CHECK_ARG(M * (long long)hidden * 2 < 0xFFFFFF00ll && aligned16(x) && aligned16(w1) && aligned16(w2) && aligned16(h) && aligned16(y));
    constexpr int lds = 2 * M0_N * 192 + M0_N * 4;
    static std::atomic<unsigned long long> configured{0};
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&mlp0_fwd_kernel), lds, configured)) return rc;
    hipLaunchKernelGGL(mlp0_fwd_kernel, dim3(mlp0_grid(M)), dim3(M0_THREADS), lds, (hipStream_t)stream, x, w1, b1, w2, h, y, (int)M);
    PSWIN_LAUNCH_RET();
}

int pswin_fc1_gelu_partial_rows(long long M) {
    if (M <= 0 || M > 0x7fffffffll) return PSWIN_ERR_ARG;
    const int ntiles = (int)((M + 15) / 16);           // RT = 1 for the fused kernels
    const int grid = (ntiles + 3) / 4;
    return grid > MAX_GRID ? MAX_GRID : grid;
}

}  // extern "C"
