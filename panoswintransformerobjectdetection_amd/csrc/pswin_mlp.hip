// Bias + exact GELU between the two Linear layers of Mlp (HOT:44-61: fc1 -> nn.GELU -> fc2), gfx950.
//
//   forward   h = gelu(y + bias)            y = x W1^T (the GEMM runs without a bias epilogue)
//   backward  dy = dh * gelu'(y + bias),    dbias = sum_rows dy
//
// One pass each over the [M, 4C] hidden activation, the largest tensor of a block.  The backward pass replaces
// the framework's GELU-backward kernel AND the separate column-sum pass for fc1's bias gradient (which re-read the
// 201 MB dy of a stage-0 block): the per-channel sums are accumulated from the values in flight (f32, before the
// bf16 rounding of dy), reduced per block through LDS and finished by the fixed-order column sum.
// A thread owns one 16-byte channel group; a block covers GW groups x RPS rows per step and strides over the rows.
#include "pswin_common.hpp"
#include "pswin_gelu.hpp"

using namespace pswin;

namespace {

constexpr int THREADS = 256;
constexpr int BWD_REP = 4;
int g_unr_fwd = 2, g_unr_bwd = 4;             // row steps per block (pswin_bias_gelu_tune)

template <int DT>
struct Vec {
    static constexpr int VE = (DT == PSWIN_BF16) ? 8 : 4;
};

template <int DT>
__device__ inline void load_vec(const void* base, size_t elem_off, float (&v)[Vec<DT>::VE]) {
    if constexpr (DT == PSWIN_BF16) {
        const u32x4 raw = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(base) + elem_off);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = __builtin_bit_cast(float, raw[e] << 16);
            v[2 * e + 1] = __builtin_bit_cast(float, raw[e] & 0xffff0000u);
        }
    } else {
        const f32x4 r = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + elem_off);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = r[e];
    }
}

template <int DT>
__device__ inline void store_vec(void* base, size_t elem_off, const float (&v)[Vec<DT>::VE]) {
    if constexpr (DT == PSWIN_BF16) {
        u32x4 raw;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            raw[e] = pack2_bf16(v[2 * e], v[2 * e + 1]);
        *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(base) + elem_off) = raw;
    } else {
        f32x4 r = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + elem_off) = r;
    }
}

// grid: (row blocks, column slabs of gw groups); rps = THREADS / gw rows per step.  Streaming kernel, one short trip per
// thread: the block owns UNR * rps CONSECUTIVE rows (a contiguous stretch of memory) and every thread issues its UNR
// 16-byte loads (2 UNR backward) before the math.  No grid-stride loop: plain streaming kernels with a huge grid reach
// ~6 TB/s on MI355X, persistent strided ones 4-5 TB/s (measured on this op).  32-bit element offsets (checked by the host).
template <int DT, bool BWD, int UNR>
__global__ __launch_bounds__(THREADS) void bias_gelu_kernel(const void* __restrict__ a, const void* __restrict__ y,
                                                            const float* __restrict__ bias, void* __restrict__ out,
                                                            float* __restrict__ partial, int M, int N, int gw, int rps) {
    constexpr int VE = Vec<DT>::VE;
    __shared__ float red[BWD ? THREADS * VE : 1];
    const unsigned rl = threadIdx.x / (unsigned)gw, cg = threadIdx.x - rl * gw;
    const unsigned col = (blockIdx.y * gw + cg) * VE;
    float b[VE], acc[VE];
    if (bias) {
#pragma unroll
        for (int v4 = 0; v4 < VE / 4; ++v4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(bias + col + 4 * v4);
#pragma unroll
            for (int e = 0; e < 4; ++e) b[4 * v4 + e] = t[e];
        }
    } else {
#pragma unroll
        for (int e = 0; e < VE; ++e) b[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
    // backward: REP batches of UNR row steps per block, so that the block's column-sum epilogue (LDS combine + one
    // partial row) is paid once per REP * UNR * rps rows and the partial rows to be summed later are REP x fewer
    constexpr int REP = BWD ? BWD_REP : 1;
    if (rl < (unsigned)rps)
      for (int rep = 0; rep < REP; ++rep) {
        const unsigned row0 = (blockIdx.x * REP + rep) * (UNR * rps) + rl;
        float yv[UNR][VE], dh[BWD ? UNR : 1][VE];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const unsigned row = row0 + u * rps;
            if (row < (unsigned)M) {
                load_vec<DT>(y, row * (unsigned)N + col, yv[u]);
                if constexpr (BWD) load_vec<DT>(a, row * (unsigned)N + col, dh[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const unsigned row = row0 + u * rps;
            if (row < (unsigned)M) {
                float o[VE];
                if constexpr (BWD) {
#pragma unroll
                    for (int e = 0; e < VE; ++e) {
                        o[e] = dh[u][e] * gelu_grad_f(yv[u][e] + b[e]);
                        acc[e] += o[e];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < VE; ++e) o[e] = gelu_f(yv[u][e] + b[e]);
                }
                store_vec<DT>(out, row * (unsigned)N + col, o);
            }
        }
    }
    if constexpr (BWD) {
#pragma unroll
        for (int e = 0; e < VE; ++e) red[threadIdx.x * VE + e] = acc[e];
        __syncthreads();
        if (rl == 0) {
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                float s = 0.f;
                for (int q = 0; q < rps; ++q) s += red[(q * gw + cg) * VE + e];
                partial[(size_t)blockIdx.x * N + col + e] = s;
            }
        }
    }
}

// column groups per block: the largest divisor of the groups per row that is <= 64
inline int pick_gw(int groups) {
    for (int d = 64; d >= 1; --d)
        if (groups % d == 0) return d;
    return 1;
}
inline long long row_blocks(long long M, int rps, int unr) {
    const long long per = (long long)rps * unr;
    return (M + per - 1) / per;
}

template <bool BWD, int UNR>
int launch_u(const void* a, const void* y, int dtype, const float* bias, void* out, float* partial, long long M, int N,
             hipStream_t st, int* blocks_out) {
    const int ve = dtype == PSWIN_BF16 ? 8 : 4;
    const int groups = N / ve, gw = pick_gw(groups);
    const int rps = THREADS / gw;
    const long long bx = row_blocks(M, rps, UNR * (BWD ? BWD_REP : 1));
    if (bx > 0x7fffffffll || M * N >= 0x7fffffffll) return PSWIN_ERR_ARG;
    if (blocks_out) *blocks_out = (int)bx;
    dim3 grid((unsigned)bx, groups / gw);
    if (dtype == PSWIN_BF16)
        hipLaunchKernelGGL((bias_gelu_kernel<PSWIN_BF16, BWD, UNR>), grid, dim3(THREADS), 0, st, a, y, bias, out, partial, (int)M, N, gw, rps);
    else
        hipLaunchKernelGGL((bias_gelu_kernel<PSWIN_F32, BWD, UNR>), grid, dim3(THREADS), 0, st, a, y, bias, out, partial, (int)M, N, gw, rps);
    PSWIN_LAUNCH_RET();
}

template <bool BWD>
int launch(const void* a, const void* y, int dtype, const float* bias, void* out, float* partial, long long M, int N,
           hipStream_t st, int* blocks_out) {
    switch (BWD ? g_unr_bwd : g_unr_fwd) {
        case 1: return launch_u<BWD, 1>(a, y, dtype, bias, out, partial, M, N, st, blocks_out);
        case 2: return launch_u<BWD, 2>(a, y, dtype, bias, out, partial, M, N, st, blocks_out);
        default: return launch_u<BWD, 4>(a, y, dtype, bias, out, partial, M, N, st, blocks_out);
    }
}

}  // namespace

extern "C" {

int pswin_bias_gelu_fwd(const void* y, int dtype, const float* bias, void* h, long long M, int N, void* stream) {
    PSWIN_CHECK_ARG(y && h && M > 0 && N > 0 && N % 8 == 0 && valid_dtype(dtype) && aligned16(y) && aligned16(h));
    return launch<false>(nullptr, y, dtype, bias, h, nullptr, M, N, (hipStream_t)stream, nullptr);
}

/* tuning hook: row steps per block (1, 2 or 4) of the forward / backward kernel */
int pswin_bias_gelu_tune(int unr_fwd, int unr_bwd) {
    if ((unr_fwd != 1 && unr_fwd != 2 && unr_fwd != 4) || (unr_bwd != 1 && unr_bwd != 2 && unr_bwd != 4)) return PSWIN_ERR_ARG;
    g_unr_fwd = unr_fwd;
    g_unr_bwd = unr_bwd;
    return PSWIN_OK;
}

int pswin_bias_gelu_workspace(long long M, int N) {
    if (M <= 0 || N <= 0 || N % 8) return PSWIN_ERR_ARG;
    // one partial row of N sums per row block; the narrowest block (64 groups) has 4 rows per step, at least 1 step
    const long long rows = row_blocks(M, THREADS / 64, 1);
    const long long n = rows * N;
    return n > 0x7fffffffll ? PSWIN_ERR_ARG : (int)n;
}

int pswin_bias_gelu_bwd(const void* dh, const void* y, int dtype, const float* bias, void* dy, float* dbias,
                        float* workspace, long long M, int N, void* stream) {
    PSWIN_CHECK_ARG(dh && y && dy && workspace && M > 0 && N > 0 && N % 8 == 0 && valid_dtype(dtype));
    PSWIN_CHECK_ARG(aligned16(dh) && aligned16(y) && aligned16(dy));
    int blocks = 0;
    int rc = launch<true>(dh, y, dtype, bias, dy, workspace, M, N, (hipStream_t)stream, &blocks);
    if (rc) return rc;
    if (dbias) launch_colsum(workspace, blocks, N, dbias, (hipStream_t)stream);   // else: partial rows only
    PSWIN_LAUNCH_RET();
}

int pswin_bias_gelu_partial_rows(long long M, int N, int dtype) {
    if (M <= 0 || N <= 0 || N % 8 || !valid_dtype(dtype)) return PSWIN_ERR_ARG;
    const int gw = pick_gw(N / (dtype == PSWIN_BF16 ? 8 : 4));
    const int unr = g_unr_bwd == 1 ? 1 : (g_unr_bwd == 2 ? 2 : 4);
    const long long bx = row_blocks(M, THREADS / gw, unr * BWD_REP);
    return bx > 0x7fffffffll ? PSWIN_ERR_ARG : (int)bx;
}

}  // extern "C"
