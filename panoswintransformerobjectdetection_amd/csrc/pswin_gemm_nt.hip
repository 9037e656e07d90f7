// Tiled bf16 GEMM for the Linear layers of stages 1-3 (gfx950):  Y[M, N] = X[M, K] . W[N, K]^T (+ bias)
//
// qkv / proj / fc1 / fc2 / reduction of PanoSwin (HOT:287, 309, 50-58, 575) at M = 4k-75k rows, K, N in 192..3072.  The
// library kernels PyTorch-ROCm selects for these shapes run at 0.4-0.8 PFLOP/s (profiles/r01: stage-2 qkv 35.7 us against a
// 9.7 us floor).  The same kernel serves the data gradient dX = dY . W through a transposed bf16 copy of the weight
// (dX[M, K] = dY[M, N] . (W^T)[K, N]^T).  bf16 operands, f32 accumulation on v_mfma_f32_16x16x32_bf16: the arithmetic of the
// library path.
//
// Design (CDNA4):
//   * macro tile BM x 192, BK = 64: every N of the model is a multiple of 192 (192 .. 3072), so no column tile is padded;
//     BM = 128 (4 waves as 2 x 2, each 64 rows x 96 columns = 24 accumulator quads) or BM = 64 for the small-M stage-3
//     shapes (twice the tiles for the 256 CUs);
//   * both operand tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, 10 instructions per wave
//     and k-step), double buffered, one barrier per k-step: the loads of step t+1 are issued right after the barrier that
//     retires step t-1's reads and fly under step t's 48 MFMAs per wave; 80 KB of LDS per workgroup -> two workgroups per
//     CU fill each other's barrier / drain bubbles;
//   * LDS rows are 128 B (64 bf16); the 16-byte chunk index is XOR-ed with (row & 7).  LDS-DMA writes lane-linear, so the
//     permutation is applied to the SOURCE address (lane i fetches chunk (i & 7) ^ (row & 7) of its row) and again on the
//     fragment reads: conflict-free ds_read_b128 for 16 consecutive rows;
//   * the product is computed transposed (A = weight rows, B = activation rows): a lane then owns 4 consecutive output
//     columns of ONE row, two column tiles are exchanged across lane groups (v_permlane16_swap) and rows leave as 16-byte
//     stores;
//   * tiles are dealt to the XCDs in contiguous chunks (blockIdx % 8 = XCD under round-robin placement; speed only) with
//     the column tile fastest, so the tiles that share an activation panel hit the same L2.
#include "pswin_common.hpp"
#include "pswin_gelu.hpp"

using namespace pswin;

namespace {

constexpr int BN = 192, BK = 64, NT_THREADS = 256;
#ifndef PSWIN_NT_PROBE
#define PSWIN_NT_PROBE 0      // tools/probe/nt_probe.hip builds ablated loops (1: no loads inside the k loop, 2: loads and barriers only)
#endif
using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ inline f32x4 mfma32(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ inline unsigned pk2(float lo, float hi) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2_t)); }
// lane (c, g) holds quads q0 = col[4g..4g+3], q1 = col[16+4g..16+4g+3] of a 32-column group of one row -> after the exchange
// 8 consecutive columns starting at 8 (g >> 1) + 16 (g & 1)
__device__ inline u32x4 pack_row8(f32x4 q0, f32x4 q1) {
    const unsigned a0 = pk2(q0[0], q0[1]), a1 = pk2(q0[2], q0[3]), b0 = pk2(q1[0], q1[1]), b1 = pk2(q1[2], q1[3]);
    const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
    return u32x4{r0[0], r1[0], r0[1], r1[1]};
}

// the same exchange on f32 quads (whole-vector bit casts: hipcc folds per-element casts of vector lanes)
__device__ inline void exchange_row8(f32x4& q0, f32x4& q1) {
    const u32x4 a = __builtin_bit_cast(u32x4, q0), b = __builtin_bit_cast(u32x4, q1);
    const auto r0 = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
    const auto r2 = __builtin_amdgcn_permlane16_swap(a[2], b[2], false, false);
    const auto r3 = __builtin_amdgcn_permlane16_swap(a[3], b[3], false, false);
    q0 = __builtin_bit_cast(f32x4, u32x4{r0[0], r1[0], r2[0], r3[0]});
    q1 = __builtin_bit_cast(f32x4, u32x4{r0[1], r1[1], r2[1], r3[1]});
}
template <int CTRL>
__device__ inline float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ inline float row16_sum(float v) {   // over the 16 lanes of a group (every lane gets the sum)
    v = dpp_add<0xB1>(v);
    v = dpp_add<0x4E>(v);
    v = dpp_add<0x141>(v);
    return dpp_add<0x140>(v);
}

typedef __attribute__((address_space(3))) void lds_void;
__device__ inline void glds16(const void* gsrc, char* lds_wave_base) {
    // 16 bytes per lane: LDS destination = wave-uniform base + lane * 16
    __builtin_amdgcn_global_load_lds(gsrc, (lds_void*)lds_wave_base, 16, 0, 0);
}

// EPI 0: y = acc (+ bias).  EPI 1 (the data gradient of fc2 fused with the backward of bias + GELU, HOT:50-58): acc = dL/dh of
// the tile, aux = the Mlp's pre-activation [M, N], bias = fc1's bias; y = acc * gelu'(aux + bias) and the per-column sums of y
// over the tile's rows (the fc1 bias gradient) go to partial[tile_m][N] -- dL/dh never exists in HBM and the separate
// bias + GELU backward pass (two reads and a write of [M, 4C]) is gone.  EPI 2 (fc1 + bias + nn.GELU forward): y = acc (the
// pre-activation WITHOUT bias, kept for the backward pass) and, to the second output `aux`, h = gelu(bf16(acc) + bias) -- the
// value the separate bias + GELU kernel computes from the stored pre-activation, bit for bit; the f32 bias comes in `partial`.
// S = LDS stages.  2: one k-step in flight under the current one, two workgroups per CU (the form every large launch runs).  4 (round 4):
// three k-steps in flight, one workgroup per CU -- for launches whose tiles fit the chip once: there a workgroup has nobody to overlap
// with, every k-step of the two-stage loop waits out a full (cold: HBM) load latency, and the launch takes ramp + K / 64 latencies
// whatever its size.  The wait in front of a step is then a COUNTED vmcnt that leaves the younger stages' LDS-DMA outstanding and the
// barrier a raw s_barrier (a __syncthreads() drains vmcnt to 0), as in pswin_gemm_tn.hip.
template <int BM, int EPI, int S = 2>
__global__ __launch_bounds__(NT_THREADS, 2) void gemm_nt_kernel(const unsigned short* __restrict__ X, const unsigned short* __restrict__ W,
                                                                 const float* __restrict__ bias, unsigned short* __restrict__ Y,
                                                                 int M, int N, int K, int tiles_m, int tiles_n,
                                                                 const unsigned short* __restrict__ aux, float* __restrict__ partial) {
    constexpr int RT = BM / 32;                       // 16-row tiles per wave (waves: 2 along M x 2 along N)
    constexpr int WROWS = BM / 2;                     // rows per wave
    constexpr int AJ = BM / 32, BJ = BN / 32;         // 8-row LDS-DMA blocks per wave and k-step (4 waves)
    constexpr int CT = BN / 32;                       // 16-column tiles per wave: 6
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(1024))) char smem[];       // S stages: [A tile | B tile]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int c = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;

    // tile of this workgroup: contiguous chunks of the tile list per XCD, column tile fastest
    const int ntiles = tiles_m * tiles_n;
    int t;
    {
        const int bid = blockIdx.x, q = ntiles / 8, r = ntiles % 8, xcd = bid % 8, loc = bid / 8;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;       // bijective for any ntiles
    }
    const int tm = t / tiles_n, tn = t - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // LDS-DMA source addressing: a wave instruction moves 8 rows x 128 B; lane i -> row i / 8, physical chunk i & 7, which
    // holds the row's logical chunk (i & 7) ^ (row & 7)
    const int lrow = lane >> 3, lch = (lane & 7) ^ (lrow & 7);                    // (row & 7) == lrow: row blocks start at multiples of 8
    const unsigned short* a_src[AJ];                                              // A: BM / 8 row blocks over the waves
    const unsigned short* b_src[BJ];                                              // B: 24 row blocks over the waves
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        int row = m0 + (wave * AJ + j) * 8 + lrow;
        row = row < M ? row : M - 1;                                              // rows past M: valid memory, results never stored
        a_src[j] = X + (size_t)row * K + 8 * lch;
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) b_src[j] = W + (size_t)(n0 + (wave * BJ + j) * 8 + lrow) * K + 8 * lch;
    auto issue = [&](int kt, int stage) {
        char* sa = smem + stage * STAGE;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int j = 0; j < AJ; ++j) glds16(a_src[j] + kt * BK, sa + (wave * AJ + j) * 1024);
#pragma unroll
        for (int j = 0; j < BJ; ++j) glds16(b_src[j] + kt * BK, sb + (wave * BJ + j) * 1024);
    };

    f32x4 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: row (16 i + c) of the wave's block, logical chunk 4 ks + g -> physical chunk ^ (row & 7); the wave's
    // row blocks start at multiples of 16, so (row & 7) = c & 7
    const int a_lane = (wm * WROWS + c) * 128, b_lane = (wn * (BN / 2) + c) * 128;
    int choff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) choff[ks] = ((4 * ks + g) ^ (c & 7)) << 4;

    // EPI 1: the tile of pre-activations the epilogue needs (12 x 16 B per lane at BM = 128) is requested during the last k-step,
    // so that it arrives under that step's MFMAs instead of in front of an idle epilogue
    const int d0 = 8 * (g >> 1) + 16 * (g & 1);
    [[maybe_unused]] u32x4 yv[CT / 2][RT];
    [[maybe_unused]] const rsrc_t as = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(aux), 0, EPI == 1 ? (int)((size_t)M * N * 2) : 0, 0x00020000);
    auto aux_off = [&](int jp, int i) {
        const unsigned row = (unsigned)(m0 + wm * WROWS + 16 * i + c);
        return row < (unsigned)M ? row * (unsigned)(N * 2) + (unsigned)((n0 + wn * (BN / 2) + 32 * jp + d0) * 2) : 0xFFFFFF00u;
    };

    const int KT = K / BK;
    constexpr int LOADS = AJ + BJ;                    // LDS-DMA instructions per wave and k-step
    static_assert(S == 2 || S == 4, "");
    static_assert(2 * LOADS <= 63, "vmcnt");
    issue(0, 0);
    if constexpr (S == 4) {
        if (1 < KT) issue(1, 1);
        if (2 < KT) issue(2, 2);
    }
    int stage = 0;
    for (int kt = 0; kt < KT; ++kt) {
        if constexpr (S == 2) {
            __syncthreads();                          // (vmcnt(0) + barrier) stage kt landed for every wave; stage kt-1 fully consumed
            if (kt + 1 < KT && PSWIN_NT_PROBE != 1) issue(kt + 1, (kt + 1) & 1);
            stage = kt & 1;
        } else {
            // this wave's loads of step kt have landed once only the younger steps' (at most two) are outstanding; the barrier says so for
            // every wave and that every wave is done reading step kt - 1, whose stage the next issue overwrites
            const int younger = KT - 1 - kt;
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            stage = kt & 3;
            if (kt + 3 < KT) issue(kt + 3, (kt + 3) & 3);
        }
        if (PSWIN_NT_PROBE == 2) continue;
        const char* sa = smem + stage * STAGE + a_lane;
        const char* sb = smem + stage * STAGE + A_BYTES + b_lane;
        // the fragments of the second 32-deep half are requested after the first four MFMAs of the first half and arrive under its
        // other twenty (two register sets): the wait in front of the first MFMA then covers the first set only
        u32x4 af[2][RT], bf[2][CT];
#pragma unroll
        for (int i = 0; i < RT; ++i) af[0][i] = *reinterpret_cast<const u32x4*>(sa + i * 2048 + choff[0]);
#pragma unroll
        for (int j = 0; j < CT; ++j) bf[0][j] = *reinterpret_cast<const u32x4*>(sb + j * 2048 + choff[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < CT; ++j) {
#pragma unroll
                for (int i = 0; i < RT; ++i) acc[i][j] = mfma32(bf[ks][j], af[ks][i], acc[i][j]);      // transposed product: rows = output columns
                if (ks == 0 && j == 0) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < RT; ++i) af[1][i] = *reinterpret_cast<const u32x4*>(sa + i * 2048 + choff[1]);
#pragma unroll
                    for (int jj = 0; jj < CT; ++jj) bf[1][jj] = *reinterpret_cast<const u32x4*>(sb + jj * 2048 + choff[1]);
                    if constexpr (EPI == 1) {
                        if (kt == KT - 1) {
#pragma unroll
                            for (int jp = 0; jp < CT / 2; ++jp)
#pragma unroll
                                for (int i2 = 0; i2 < RT; ++i2) yv[jp][i2] = __builtin_amdgcn_raw_buffer_load_b128(as, aux_off(jp, i2), 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // acc[i][j][e] = Y[row m0 + wm * BM/2 + 16 i + c][column n0 + wn * 96 + 16 j + 4 g + e]
    const rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(Y, 0, (int)((size_t)M * N * (EPI == 3 ? 4 : 2)), 0x00020000);
    if constexpr (EPI == 0 || EPI == 2 || EPI == 3) {
        const rsrc_t hs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(aux), 0, EPI == 2 ? (int)((size_t)M * N * 2) : 0, 0x00020000);
        // the bias values of this lane's columns are loaded before the first store: a load issued behind stores waits for them
        // (one in-order counter), which made every row tile of the epilogue a memory round trip
        f32x4 bq[CT / 2][2];
#pragma unroll
        for (int jp = 0; jp < CT / 2; ++jp) {
            const float* bp = EPI == 2 ? partial + n0 + wn * (BN / 2) + 32 * jp + d0 : bias + n0 + wn * (BN / 2) + 32 * jp + 4 * g;
            const bool have = EPI == 2 || bias != nullptr;
            bq[jp][0] = have ? *reinterpret_cast<const f32x4*>(bp) : f32x4{0.f, 0.f, 0.f, 0.f};
            bq[jp][1] = have ? *reinterpret_cast<const f32x4*>(bp + (EPI == 2 ? 4 : 16)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const unsigned row = (unsigned)(m0 + wm * WROWS + 16 * i + c);
            const unsigned base = row < (unsigned)M ? row * (unsigned)(N * 2) + (unsigned)((n0 + wn * (BN / 2) + d0) * 2) : 0xFFFFFF00u;
#pragma unroll
            for (int jp = 0; jp < CT / 2; ++jp) {
                f32x4 q0 = acc[i][2 * jp], q1 = acc[i][2 * jp + 1];
                if constexpr (EPI == 0 || EPI == 3) {
                    q0 += bq[jp][0];
                    q1 += bq[jp][1];
                }
                const unsigned off = base == 0xFFFFFF00u ? base : base + 64u * jp;
                if constexpr (EPI == 3) {             // f32 output (Y is float*): the same 8 consecutive columns as two 16-byte stores
                    exchange_row8(q0, q1);
                    const unsigned off4 = off == 0xFFFFFF00u ? off : 2u * off;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, q0), ys, off4, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, q1), ys, off4 == 0xFFFFFF00u ? off4 : off4 + 16u, 0, 0);
                    continue;
                }
                const u32x4 yp = pack_row8(q0, q1);
                __builtin_amdgcn_raw_buffer_store_b128(yp, ys, off, 0, 0);
                if constexpr (EPI == 2) {
                    // 8 consecutive columns n0 + wn * 96 + 32 jp + d0 .. of this lane's row: gelu(rounded pre-activation + fc1 bias)
                    const f32x4 b0 = bq[jp][0], b1 = bq[jp][1];
                    u32x4 hp;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const gelu_f32x2 hv = gelu_f2(gelu_f32x2{__builtin_bit_cast(float, yp[d] << 16) + (d < 2 ? b0[2 * d] : b1[2 * d - 4]),
                                                                 __builtin_bit_cast(float, yp[d] & 0xffff0000u) + (d < 2 ? b0[2 * d + 1] : b1[2 * d - 3])});
                        hp[d] = pk2(hv[0], hv[1]);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(hp, hs, off, 0, 0);
                }
            }
        }
    } else {
        // the pre-activations were requested in the last k-step; their offsets again for the stores
        unsigned offs[CT / 2][RT];
#pragma unroll
        for (int jp = 0; jp < CT / 2; ++jp)
#pragma unroll
            for (int i = 0; i < RT; ++i) offs[jp][i] = aux_off(jp, i);
        f32x4 bv[CT / 2][2];                          // the fc1 bias of this lane's columns, loaded before the first store: a load
#pragma unroll                                        // issued behind stores waits for them (one in-order counter)
        for (int jp = 0; jp < CT / 2; ++jp) {
            const int col8 = n0 + wn * (BN / 2) + 32 * jp + d0;
            bv[jp][0] = bias ? *reinterpret_cast<const f32x4*>(bias + col8) : f32x4{0.f, 0.f, 0.f, 0.f};
            bv[jp][1] = bias ? *reinterpret_cast<const f32x4*>(bias + col8 + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __builtin_amdgcn_sched_barrier(0);
        float csum[CT / 2][8];                        // column sums over this wave's rows: columns 32 jp + d0 + j of its 96
#pragma unroll
        for (int jp = 0; jp < CT / 2; ++jp) {
            const f32x4 b0 = bv[jp][0], b1 = bv[jp][1];
            float lsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};       // this lane's rows first, one cross-lane sum per column after
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const bool ok = offs[jp][i] != 0xFFFFFF00u;
                f32x4 q0 = acc[i][2 * jp], q1 = acc[i][2 * jp + 1];
                exchange_row8(q0, q1);                // 8 consecutive columns col8 .. col8 + 7 of this lane's row
                float v[8];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const gelu_f32x2 pre = {__builtin_bit_cast(float, yv[jp][i][d] << 16) + (d < 2 ? b0[2 * d] : b1[2 * d - 4]),
                                            __builtin_bit_cast(float, yv[jp][i][d] & 0xffff0000u) + (d < 2 ? b0[2 * d + 1] : b1[2 * d - 3])};
                    const gelu_f32x2 gg = gelu_grad_f2(pre);
                    v[2 * d] = (d < 2 ? q0[2 * d] : q1[2 * d - 4]) * gg[0];
                    v[2 * d + 1] = (d < 2 ? q0[2 * d + 1] : q1[2 * d - 3]) * gg[1];
                }
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{pk2(v[0], v[1]), pk2(v[2], v[3]), pk2(v[4], v[5]), pk2(v[6], v[7])}, ys, offs[jp][i], 0, 0);
#pragma unroll
                for (int j = 0; j < 8; ++j) lsum[j] += ok ? v[j] : 0.f;      // rows past M: clamped duplicates, not counted
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) csum[jp][j] = row16_sum(lsum[j]);
        }
        __syncthreads();                              // every wave is done with the operand stages: reuse them for the column sums
        float* cs = reinterpret_cast<float*>(smem);   // [4 waves][96]
        if (c == 0) {
#pragma unroll
            for (int jp = 0; jp < CT / 2; ++jp)
#pragma unroll
                for (int j = 0; j < 8; ++j) cs[wave * (BN / 2) + 32 * jp + d0 + j] = csum[jp][j];
        }
        __syncthreads();
        if (tid < BN) {
            const int wn2 = tid / (BN / 2), cc = tid - wn2 * (BN / 2);
            partial[(size_t)tm * N + n0 + tid] = cs[wn2 * (BN / 2) + cc] + cs[(2 + wn2) * (BN / 2) + cc];     // waves (wm = 0, 1; wn2)
        }
    }
}

template <int BM, int EPI, int S>
int launch_nt_s(const void* x, const void* w, const float* bias, void* y, int M, int N, int K, hipStream_t st, const void* aux, float* partial) {
    constexpr size_t lds = S * (size_t)(BM + BN) * BK * 2;
    static_assert(lds <= 160 * 1024, "LDS");
    static std::atomic<unsigned long long> configured{0};
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&gemm_nt_kernel<BM, EPI, S>), lds, configured)) return rc;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = N / BN;
    hipLaunchKernelGGL((gemm_nt_kernel<BM, EPI, S>), dim3(tiles_m * tiles_n), dim3(NT_THREADS), lds, st,
                       reinterpret_cast<const unsigned short*>(x), reinterpret_cast<const unsigned short*>(w), bias,
                       reinterpret_cast<unsigned short*>(y), M, N, K, tiles_m, tiles_n, reinterpret_cast<const unsigned short*>(aux), partial);
    PSWIN_LAUNCH_RET();
}

// launches of at most this many tiles take the four-stage form (one workgroup per CU); swept 0 / 256 / 512 / 1024 inside the step
// (profiles/r04_ab_runs.json: gemm_nt_four_stage_threshold_tiles)
constexpr int DEEP_TILES = 256;

template <int BM, int EPI>
int launch_nt(const void* x, const void* w, const float* bias, void* y, int M, int N, int K, hipStream_t st, const void* aux = nullptr,
              float* partial = nullptr) {
    const int tiles = ((M + BM - 1) / BM) * (N / BN);
    if (tiles <= DEEP_TILES && K / BK >= 3) return launch_nt_s<BM, EPI, 4>(x, w, bias, y, M, N, K, st, aux, partial);
    return launch_nt_s<BM, EPI, 2>(x, w, bias, y, M, N, K, st, aux, partial);
}

}  // namespace

extern "C" {

int pswin_gemm_nt_supported(long long M, int K, int N) {
    return M >= 64 && M * (long long)(K > N ? K : N) * 2 < 0xFFFFFF00ll && K >= 64 && K % 64 == 0 && N >= 192 && N % 192 == 0;
}

/* tile_m: 0 = choose (128-row tiles unless that leaves the 256 CUs short of two rounds of tiles), or 64 / 96 / 128 */
int pswin_gemm_nt(const void* x, const void* w, const float* bias, void* y, long long M, int K, int N, int tile_m, void* stream) {
    PSWIN_CHECK_ARG(x && w && y && pswin_gemm_nt_supported(M, K, N) && (tile_m == 0 || tile_m == 64 || tile_m == 96 || tile_m == 128));
    PSWIN_CHECK_ARG(aligned16(x) && aligned16(w) && aligned16(y) && aligned16(bias));
    const int m = (int)M;
    if (tile_m == 0) tile_m = ((long long)((m + 127) / 128) * (N / BN) >= 512) ? 128 : 64;
    if (tile_m == 128) return launch_nt<128, 0>(x, w, bias, y, m, N, K, (hipStream_t)stream);
    // 96-row tiles (plain epilogue only): for row counts whose 64-row tiles need a second, mostly empty round of the 512 tile slots
    if (tile_m == 96) return launch_nt<96, 0>(x, w, bias, y, m, N, K, (hipStream_t)stream);
    return launch_nt<64, 0>(x, w, bias, y, m, N, K, (hipStream_t)stream);
}

/* the same product written as f32 (y: f32 [M, N]): for a Linear whose result joins the fp32 residual stream (PatchMerging.reduction) */
int pswin_gemm_nt_f32(const void* x, const void* w, const float* bias, float* y, long long M, int K, int N, int tile_m, void* stream) {
    PSWIN_CHECK_ARG(x && w && y && pswin_gemm_nt_supported(M, K, N) && M * (long long)N * 4 < 0xFFFFFF00ll &&
                    (tile_m == 0 || tile_m == 64 || tile_m == 128));
    PSWIN_CHECK_ARG(aligned16(x) && aligned16(w) && aligned16(y) && aligned16(bias));
    const int m = (int)M;
    if (tile_m == 0) tile_m = ((long long)((m + 127) / 128) * (N / BN) >= 512) ? 128 : 64;
    if (tile_m == 128) return launch_nt<128, 3>(x, w, bias, y, m, N, K, (hipStream_t)stream);
    return launch_nt<64, 3>(x, w, bias, y, m, N, K, (hipStream_t)stream);
}

int pswin_gemm_nt_partial_rows(long long M, int tile_m) {
    return (M > 0 && (tile_m == 64 || tile_m == 128)) ? (int)((M + tile_m - 1) / tile_m) : PSWIN_ERR_ARG;
}

int pswin_gemm_nt_gelu_fwd(const void* x, const void* w, const float* bias, void* pre, void* h, long long M, int K, int N, int tile_m,
                           void* stream) {
    PSWIN_CHECK_ARG(x && w && bias && pre && h && pswin_gemm_nt_supported(M, K, N) && (tile_m == 64 || tile_m == 128));
    PSWIN_CHECK_ARG(aligned16(x) && aligned16(w) && aligned16(pre) && aligned16(h) && aligned16(bias));
    float* b = const_cast<float*>(bias);
    if (tile_m == 128) return launch_nt<128, 2>(x, w, nullptr, pre, (int)M, N, K, (hipStream_t)stream, h, b);
    return launch_nt<64, 2>(x, w, nullptr, pre, (int)M, N, K, (hipStream_t)stream, h, b);
}

int pswin_gemm_nt_gelu_bwd(const void* dy, const void* w_t, const void* pre, const float* bias, void* dpre, float* partial, long long M,
                           int K, int N, int tile_m, void* stream) {
    PSWIN_CHECK_ARG(dy && w_t && pre && dpre && partial && pswin_gemm_nt_supported(M, K, N) && (tile_m == 64 || tile_m == 128));
    PSWIN_CHECK_ARG(aligned16(dy) && aligned16(w_t) && aligned16(pre) && aligned16(dpre) && aligned16(bias) && aligned16(partial));
    if (tile_m == 128) return launch_nt<128, 1>(dy, w_t, bias, dpre, (int)M, N, K, (hipStream_t)stream, pre, partial);
    return launch_nt<64, 1>(dy, w_t, bias, dpre, (int)M, N, K, (hipStream_t)stream, pre, partial);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// Batched bf16 transposes: the [K][N] copies of the Linear weights that turn the data gradient dX = dY . W into the same
// "both operands contraction-contiguous" product as the forward pass.  One launch per step for all layers (the job table
// travels in the kernel arguments, as pswin_reduce_jobs).
// ---------------------------------------------------------------------------------------------------------------------
namespace {

constexpr int TJ_MAX = 64;
struct TJob {
    const unsigned short* src;
    unsigned short* dst;
    int rows, cols, first_block, tiles_c;
};
struct TBatch {
    TJob job[TJ_MAX];
    int n;
};

__global__ __launch_bounds__(256) void transpose_jobs_kernel(const TBatch b) {
    __shared__ unsigned short tile[64][66];
    int lo = 0, hi = b.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (b.job[mid].first_block <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const TJob& j = b.job[lo];
    const int tl = (int)blockIdx.x - j.first_block, tr = tl / j.tiles_c, tc = tl - tr * j.tiles_c;
    const int r0 = tr * 64, c0 = tc * 64;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8 threads, 2 elements (4 B) per thread and row
#pragma unroll
    for (int rr = ty; rr < 64; rr += 8) {
        const unsigned v = *reinterpret_cast<const unsigned*>(j.src + (size_t)(r0 + rr) * j.cols + c0 + 2 * tx);
        tile[rr][2 * tx] = (unsigned short)(v & 0xffffu);
        tile[rr][2 * tx + 1] = (unsigned short)(v >> 16);
    }
    __syncthreads();
#pragma unroll
    for (int cc = ty; cc < 64; cc += 8) {
        const unsigned v = (unsigned)tile[2 * tx][cc] | ((unsigned)tile[2 * tx + 1][cc] << 16);
        *reinterpret_cast<unsigned*>(j.dst + (size_t)(c0 + cc) * j.rows + r0 + 2 * tx) = v;
    }
}

}  // namespace

extern "C" int pswin_transpose_jobs(const pswin_transpose_job* jobs, int n_jobs, void* stream) {
    PSWIN_CHECK_ARG(jobs && n_jobs > 0);
    for (int j = 0; j < n_jobs; ++j)
        PSWIN_CHECK_ARG(jobs[j].src && jobs[j].dst && jobs[j].rows > 0 && jobs[j].cols > 0 && jobs[j].rows % 64 == 0 && jobs[j].cols % 64 == 0);
    for (int at = 0; at < n_jobs; at += TJ_MAX) {
        TBatch b;
        b.n = n_jobs - at < TJ_MAX ? n_jobs - at : TJ_MAX;
        long long blocks = 0;
        for (int j = 0; j < b.n; ++j) {
            const pswin_transpose_job& q = jobs[at + j];
            TJob& t = b.job[j];
            t.src = reinterpret_cast<const unsigned short*>(q.src);
            t.dst = reinterpret_cast<unsigned short*>(q.dst);
            t.rows = q.rows;
            t.cols = q.cols;
            t.first_block = (int)blocks;
            t.tiles_c = q.cols / 64;
            blocks += (long long)(q.rows / 64) * (q.cols / 64);
        }
        PSWIN_CHECK_ARG(blocks < 0x7fffffffll);
        hipLaunchKernelGGL(transpose_jobs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, b);
    }
    PSWIN_LAUNCH_RET();
}
