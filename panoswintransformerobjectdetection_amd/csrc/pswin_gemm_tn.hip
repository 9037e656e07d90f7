// Weight-gradient GEMM of the Linear layers of stages 1-3 (gfx950):  dW[N, K] = dY[M, N]^T . X[M, K]
//
// The contraction runs over the M = 4k-75k token rows, i.e. over the STRIDED index of both row-major operands ("TN").
// PyTorch-ROCm serves it as a batched split-K library GEMM at 0.33 of its floor (2.05 ms of a 12.3 ms PanoSwin-T step,
// profiles/r02).  Here: 64-row slabs of dY and X go global -> LDS by LDS-DMA (buffer_load ... lds: rows past M read as
// zeros through the buffer range check), and BOTH MFMA operands are read from the row-major LDS panels TRANSPOSED with
// ds_read_b64_tr_b16 (two 8-byte reads = the 8 consecutive rows m of one column that a lane contributes to a 32-deep step).
//
// Design (CDNA4):
//   * LDS panels of [64 rows m][64 columns] bf16 (128-byte rows); the 32-byte block index of a row is XOR-ed with
//     ((m >> 1) & 1) | (((m >> 3) & 1) << 1): the 8 (row, block) pieces a half-wave's transposed read touches then fall on 8
//     different 32-byte slots of the 256-byte bank row (conflict-free); LDS-DMA writes lane-linear, so the permutation is
//     applied to the source column of each lane;
//   * macro tile 128 (k) x 192 (n) or 192 x 128 outputs (2 + 3 panels per stage, 40 KB; double buffered: two workgroups per
//     CU), 4 waves as 2 x 2, 24 accumulator quads per wave, 48 MFMAs per wave and 64-row step -- the geometry of
//     pswin_gemm_nt; every N, K of the model is a multiple of 192 and all but the stage-1 qkv / proj pair have one of the
//     two a multiple of 128;
//   * the M rows are split over `splits` workgroups per output tile (the tiles alone are 8-96 workgroups); each split writes
//     its f32 partial tile [split][N][K] once with 16-byte stores (the product is computed as X^T . dY so that a lane owns 4
//     consecutive k of one row n) and the caller sums the splits in its fixed-order grouped reduction (pswin_reduce_jobs)
//     straight into the flat gradient buffer -- no atomics, bitwise reproducible.
#include "pswin_common.hpp"

using namespace pswin;

namespace {

constexpr int TN_THREADS = 256, MSTEP = 64, PANEL = 64 * 64 * 2;          // one [64][64] bf16 panel: 8 KB
using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
typedef __attribute__((ext_vector_type(8))) short s16x8;

__device__ inline f32x4 mfma32(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// 8 consecutive rows (r .. r+3 at `p`, r+4 .. r+7 at p + 4 rows) of the lane's column -> one operand fragment
__device__ inline bf16x8 read_tr8(const char* p) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 4 * 128));
    const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, both);
}

// PA / PB: 64-column panels of X (k side) / dY (n side) per macro tile
template <int PA, int PB>
__global__ __launch_bounds__(TN_THREADS, 2) void gemm_tn_kernel(const unsigned short* __restrict__ DY, const unsigned short* __restrict__ X,
                                                                 float* __restrict__ P, int M, int N, int K, int tiles_k, int tiles_n,
                                                                 int rows_per_split) {
    constexpr int IA = 2 * PA, JB = 2 * PB;           // 16-wide tiles per wave along k / n (waves: 2 x 2)
    constexpr int STAGE = (PA + PB) * PANEL;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int c = lane & 15, g = lane >> 4;
    const int wk = wave >> 1, wn = wave & 1;

    // (split, tile) of this workgroup: contiguous chunks per XCD, tile fastest: the tiles of one split share its rows in L2
    const int ntiles = tiles_k * tiles_n, nwg = gridDim.x;
    int t;
    {
        const int bid = blockIdx.x, q = nwg / 8, r = nwg % 8, xcd = bid % 8, loc = bid / 8;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int split = t / ntiles, tile = t - split * ntiles;
    const int tk = tile / tiles_n, tn = tile - tk * tiles_n;
    const int k0 = tk * 64 * PA, n0 = tn * 64 * PB;
    const int m_begin = split * rows_per_split;
    const int steps = rows_per_split / MSTEP;

    const rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(X), 0, (int)((size_t)M * K * 2), 0x00020000);
    const rsrc_t ds = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(DY), 0, (int)((size_t)M * N * 2), 0x00020000);
    // LDS-DMA: one wave instruction = 8 rows x 128 B of a panel; lane i -> row i / 8, physical 16-byte chunk i & 7 = half
    // (i & 1) of physical 32-byte block (i & 7) >> 1, which holds logical block ^ f(row)
    const int lr = lane >> 3, pch = lane & 7;
    auto issue = [&](int step, int stage) {
        char* base = smem + stage * STAGE;
        const int m = m_begin + step * MSTEP;
#pragma unroll
        for (int j = 0; j < (PA + PB) * 2; ++j) {                      // (PA + PB) * 8 row blocks over 4 waves
            const int blk_all = wave * (PA + PB) * 2 + j;               // 0 .. (PA + PB) * 8 - 1
            const int panel = blk_all >> 3, blk = blk_all & 7;
            const int row = 8 * blk + lr;
            const int f = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
            const int col = (((pch >> 1) ^ f) << 4) + ((pch & 1) << 3);                           // logical column inside the panel
            char* dst = base + panel * PANEL + blk * 1024;
            if (panel < PA) {
                const unsigned off = (unsigned)(m + row) * (unsigned)(K * 2) + (unsigned)((k0 + 64 * panel + col) * 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xs, (lds_void*)dst, 16, off, 0, 0, 0);
            } else {
                const unsigned off = (unsigned)(m + row) * (unsigned)(N * 2) + (unsigned)((n0 + 64 * (panel - PA) + col) * 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ds, (lds_void*)dst, 16, off, 0, 0, 0);
            }
        }
    };

    f32x4 acc[IA][JB];
#pragma unroll
    for (int i = 0; i < IA; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposed fragment reads: lane (c = 4 q + p, g) addresses row (32 ks + 8 g + q), 8-byte piece p of the 16-column block
    // cb of a panel; the block's swizzle f = ((q >> 1) & 1) | ((g & 1) << 1) is a lane constant
    const int q = c >> 2, p8 = c & 3;
    const int fl = ((q >> 1) & 1) | ((g & 1) << 1);
    const int lane_row = (8 * g + q) * 128 + p8 * 8;
    int xoff[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) xoff[cb] = ((cb ^ fl) << 5) + lane_row;

    issue(0, 0);
    for (int st = 0; st < steps; ++st) {
        __syncthreads();                              // (vmcnt(0) + barrier) slab st landed for every wave; slab st-1 fully consumed
        if (st + 1 < steps) issue(st + 1, (st + 1) & 1);
        const char* sa = smem + (st & 1) * STAGE;                      // X panels
        const char* sb = sa + PA * PANEL;                              // dY panels
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[IA], bfr[JB];
#pragma unroll
            for (int i = 0; i < IA; ++i) {
                const int col0 = wk * (32 * PA) + 16 * i;               // column inside the k extent of the tile
                af[i] = read_tr8(sa + (col0 >> 6) * PANEL + ks * 32 * 128 + xoff[(col0 >> 4) & 3]);
            }
#pragma unroll
            for (int j = 0; j < JB; ++j) {
                const int col0 = wn * (32 * PB) + 16 * j;
                bfr[j] = read_tr8(sb + (col0 >> 6) * PANEL + ks * 32 * 128 + xoff[(col0 >> 4) & 3]);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < JB; ++j)
#pragma unroll
                for (int i = 0; i < IA; ++i) acc[i][j] = mfma32(af[i], bfr[j], acc[i][j]);      // rows = k, columns = n
            __builtin_amdgcn_s_setprio(0);
        }
    }

    // acc[i][j][e] = partial dW[n = n0 + wn * 32 PB + 16 j + c][k = k0 + wk * 32 PA + 16 i + 4 g + e]
    float* out = P + (size_t)split * N * K;
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const int n = n0 + wn * (32 * PB) + 16 * j + c;
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const int k = k0 + wk * (32 * PA) + 16 * i + 4 * g;
            *reinterpret_cast<f32x4*>(out + (size_t)n * K + k) = acc[i][j];
        }
    }
}

template <int PA, int PB>
int launch_tn(const void* dy, const void* x, float* partial, int M, int N, int K, int splits, hipStream_t st) {
    constexpr size_t lds = 2 * (size_t)(PA + PB) * PANEL;
    static std::atomic<unsigned long long> configured{0};
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&gemm_tn_kernel<PA, PB>), lds, configured)) return rc;
    const int tiles_k = K / (64 * PA), tiles_n = N / (64 * PB);
    const int rows_per_split = ((M + splits - 1) / splits + MSTEP - 1) / MSTEP * MSTEP;
    hipLaunchKernelGGL((gemm_tn_kernel<PA, PB>), dim3(tiles_k * tiles_n * splits), dim3(TN_THREADS), lds, st,
                       reinterpret_cast<const unsigned short*>(dy), reinterpret_cast<const unsigned short*>(x), partial, M, N, K, tiles_k,
                       tiles_n, rows_per_split);
    PSWIN_LAUNCH_RET();
}

// which macro tile: 1 = 128 (k) x 192 (n), 2 = 192 (k) x 128 (n), 0 = neither fits
inline int tn_shape(int N, int K) {
    if (K % 128 == 0 && N % 192 == 0) return 1;
    if (K % 192 == 0 && N % 128 == 0) return 2;
    return 0;
}

}  // namespace

extern "C" {

int pswin_gemm_tn_supported(long long M, int N, int K) {
    return M >= 64 && M * (long long)(K > N ? K : N) * 2 < 0x7fffffffll && N >= 128 && K >= 128 && tn_shape(N, K) != 0;
}

/* number of row splits: about 384 workgroups in all (measured optimum on MI355X for every stage 1-3 shape of PanoSwin-T,
 * profiles/r02_gemm_tn_split_sweep.txt: fewer leave CUs idle, more pay for their extra f32 partial slabs) while every split
 * keeps >= 256 rows */
int pswin_gemm_tn_splits(long long M, int N, int K) {
    if (!pswin_gemm_tn_supported(M, N, K)) return PSWIN_ERR_ARG;
    const int tiles = (N / 64) * (K / 64) / 6;
    int s = (384 + tiles - 1) / tiles;
    const int smax = (int)(M / 256) > 0 ? (int)(M / 256) : 1;
    if (s > smax) s = smax;
    return s < 1 ? 1 : s;
}

int pswin_gemm_tn(const void* dy, const void* x, float* partial, long long M, int N, int K, int splits, void* stream) {
    PSWIN_CHECK_ARG(dy && x && partial && pswin_gemm_tn_supported(M, N, K) && splits >= 1 && splits <= M / 64);
    PSWIN_CHECK_ARG(aligned16(dy) && aligned16(x) && aligned16(partial));
    const int shape = tn_shape(N, K);
    if (shape == 1) return launch_tn<2, 3>(dy, x, partial, (int)M, N, K, splits, (hipStream_t)stream);
    return launch_tn<3, 2>(dy, x, partial, (int)M, N, K, splits, (hipStream_t)stream);
}

}  // extern "C"
