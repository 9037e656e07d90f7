// BatchNorm2d (batch statistics) fused with ReLU for the PatchEmbed stem, channels-last (gfx950).
//
// PatchEmbed (HOT:742-750) is Conv3x3 -> BN -> ReLU -> Conv3x3 -> BN -> ReLU -> Conv4x4/s4 at FULL input resolution:
// the two BN+ReLU pairs touch the largest activations of the whole network (268 MB and 537 MB in bf16 at B = 8,
// 512x1024).  Framework kernels spend 3 passes + a ReLU pass forward and 3 passes + a ReLU-backward pass backward on
// them; here it is the minimum for training-mode BN: forward = one statistics read + one read/write apply pass,
// backward = one reduction read (dz, y) + one read/write pass (dz, y -> dy), ReLU folded into both.
// HOT = mmdet/models/backbones/simple_panoswin_transformer.py of the reference.
//
// Layout: y is [M, C] rows (M = N*H*W pixels of an NHWC tensor, C = 32 / 64 / ... channels, C % 8 == 0).
// A thread owns one 16-byte channel group (8 bf16 / 4 f32) of a row; a 256-thread block covers 256 / (C / VE) rows per
// step and strides over the rows; per-channel partial sums stay in registers, are combined across the block's row
// lanes through LDS and written as one partial row per block; a fixed-order column sum finishes them.
#include "pswin_common.hpp"

using namespace pswin;

namespace {

constexpr int THREADS = 256;
constexpr int MAX_BLOCKS = 1024;

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

// Two per-channel sums over the rows.  MODE 0 (forward stats): (sum y, sum y^2).
// MODE 1 (backward): g = dz * [relu input > 0]; (sum g, sum g * xhat).   partial: [gridDim.x][2][C]
template <int DT, int MODE>
__global__ __launch_bounds__(THREADS) void bn_reduce_kernel(const void* __restrict__ y, const void* __restrict__ dz,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, long long M, int C,
                                                            float* __restrict__ partial) {
    constexpr int VE = Vec<DT>::VE;
    __shared__ float red[2][THREADS * VE];
    const int vpr = C / VE;                      // channel groups per row
    const int rpi = THREADS / vpr;               // rows per block step
    const int cg = threadIdx.x % vpr, rl = threadIdx.x / vpr;
    float a0[VE], a1[VE], sc[VE], sh[VE], mu[VE], rs[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        a0[e] = a1[e] = 0.f;
        if (MODE == 1) {
            sc[e] = scale[cg * VE + e]; sh[e] = shift[cg * VE + e];
            mu[e] = mean[cg * VE + e]; rs[e] = rstd[cg * VE + e];
        }
    }
    if (rl < rpi) {
        for (long long r = (long long)blockIdx.x * rpi + rl; r < M; r += (long long)gridDim.x * rpi) {
            float v[VE];
            load_vec<DT>(y, (size_t)r * C + (size_t)cg * VE, v);
            if constexpr (MODE == 0) {
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    a0[e] += v[e];
                    a1[e] = __builtin_fmaf(v[e], v[e], a1[e]);
                }
            } else {
                float g[VE];
                load_vec<DT>(dz, (size_t)r * C + (size_t)cg * VE, g);
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    const float gz = (__builtin_fmaf(v[e], sc[e], sh[e]) > 0.f) ? g[e] : 0.f;
                    a0[e] += gz;
                    a1[e] = __builtin_fmaf(gz, (v[e] - mu[e]) * rs[e], a1[e]);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        red[0][threadIdx.x * VE + e] = a0[e];
        red[1][threadIdx.x * VE + e] = a1[e];
    }
    __syncthreads();
    if (rl == 0) {
        float* out = partial + (size_t)blockIdx.x * 2 * C;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            float s0 = 0.f, s1 = 0.f;
            for (int q = 0; q < rpi; ++q) {
                s0 += red[0][(q * vpr + cg) * VE + e];
                s1 += red[1][(q * vpr + cg) * VE + e];
            }
            out[cg * VE + e] = s0;
            out[C + cg * VE + e] = s1;
        }
    }
}

// sums: [2][C] = (sum y, sum y^2).  Writes mean, rstd, scale = gamma*rstd, shift = beta - mean*scale; updates the
// running statistics like nn.BatchNorm2d (momentum, unbiased variance).
__global__ void bn_finalize_kernel(const float* __restrict__ sums, long long M, int C, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, const float* __restrict__ mean_offset, float eps,
                                   float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ scale,
                                   float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = (double)sums[c] / (double)M;
    double var = (double)sums[C + c] / (double)M - m * m;      // fp64 for the cancellation in E[y^2] - E[y]^2
    if (var < 0.0) var = 0.0;
    const float r = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)m;
    rstd[c] = r;
    const float s = gamma[c] * r;
    scale[c] = s;
    shift[c] = beta[c] - (float)m * s;
    if (running_mean) {
        // a per-channel constant added in front of BN (the convolution bias) cancels in the output and only moves
        // the tracked mean: it is applied here instead of in a pass over the activation
        const float tracked = (float)m + (mean_offset ? mean_offset[c] : 0.f);
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * tracked;
        const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// eval mode: statistics are the running ones
__global__ void bn_eval_params_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ mean_offset, float eps,
                                      const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                      float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ scale,
                                      float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float r = 1.0f / sqrtf(running_var[c] + eps);
    const float m = running_mean[c] - (mean_offset ? mean_offset[c] : 0.f);   // y here excludes the offset
    mean[c] = m;
    rstd[c] = r;
    scale[c] = gamma[c] * r;
    shift[c] = beta[c] - m * gamma[c] * r;
}

// scale = gamma * rstd, shift = beta - mean * scale from saved statistics (backward pass)
__global__ void bn_scale_shift_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                      float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s = gamma[c] * rstd[c];
    scale[c] = s;
    shift[c] = beta[c] - mean[c] * s;
}

// z = relu(y * scale + shift)
template <int DT>
__global__ __launch_bounds__(THREADS) void bn_relu_apply_kernel(const void* __restrict__ y,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift, void* __restrict__ z,
                                                                long long n_vec, int C) {
    constexpr int VE = Vec<DT>::VE;
    const int vpr = C / VE;
    for (long long i = (long long)blockIdx.x * THREADS + threadIdx.x; i < n_vec; i += (long long)gridDim.x * THREADS) {
        const int cg = (int)(i % vpr);
        float v[VE];
        load_vec<DT>(y, (size_t)i * VE, v);
#pragma unroll
        for (int e = 0; e < VE; ++e) v[e] = fmaxf(__builtin_fmaf(v[e], scale[cg * VE + e], shift[cg * VE + e]), 0.f);
        store_vec<DT>(z, (size_t)i * VE, v);
    }
}

// sums: [2][C] = (sum g, sum g*xhat) -> dgamma = sum g*xhat, dbeta = sum g, and the two means the apply pass needs
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ sums, long long M, int C, int train,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ mg, float* __restrict__ mgx) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    dbeta[c] = sums[c];
    dgamma[c] = sums[C + c];
    mg[c] = train ? sums[c] / (float)M : 0.f;          // eval mode: statistics are constants, no mean terms
    mgx[c] = train ? sums[C + c] / (float)M : 0.f;
}

// dy = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dz * [relu input > 0]
template <int DT>
__global__ __launch_bounds__(THREADS) void bn_relu_bwd_apply_kernel(const void* __restrict__ dz,
                                                                    const void* __restrict__ y,
                                                                    const float* __restrict__ scale,
                                                                    const float* __restrict__ shift,
                                                                    const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd,
                                                                    const float* __restrict__ mg,
                                                                    const float* __restrict__ mgx, void* __restrict__ dy,
                                                                    long long n_vec, int C) {
    constexpr int VE = Vec<DT>::VE;
    const int vpr = C / VE;
    for (long long i = (long long)blockIdx.x * THREADS + threadIdx.x; i < n_vec; i += (long long)gridDim.x * THREADS) {
        const int cg = (int)(i % vpr);
        float v[VE], g[VE];
        load_vec<DT>(y, (size_t)i * VE, v);
        load_vec<DT>(dz, (size_t)i * VE, g);
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            const int c = cg * VE + e;
            const float gz = (__builtin_fmaf(v[e], scale[c], shift[c]) > 0.f) ? g[e] : 0.f;
            const float xh = (v[e] - mean[c]) * rstd[c];
            v[e] = scale[c] * (gz - mg[c] - xh * mgx[c]);
        }
        store_vec<DT>(dy, (size_t)i * VE, v);
    }
}

inline int reduce_blocks(long long M, int C, int dtype) {
    const int ve = dtype == PSWIN_BF16 ? 8 : 4;
    const int rpi = THREADS / (C / ve);
    long long nb = (M + (long long)rpi * 16 - 1) / ((long long)rpi * 16);
    if (nb > MAX_BLOCKS) nb = MAX_BLOCKS;
    return nb < 1 ? 1 : (int)nb;
}

inline int apply_blocks(long long n_vec) {
    long long nb = (n_vec + THREADS - 1) / THREADS;
    return (int)(nb > 8192 ? 8192 : nb);
}

inline bool bn_args_ok(long long M, int C, int dtype) {
    const int ve = dtype == PSWIN_BF16 ? 8 : 4;
    return M > 0 && C >= ve && C % 8 == 0 && C / ve <= THREADS && valid_dtype(dtype);
}

// workspace layout (floats): partial [MAX_BLOCKS][2C] | sums [2C] | scale [C] | shift [C] | mg [C] | mgx [C]
struct Ws {
    float *partial, *sums, *scale, *shift, *mg, *mgx;
};
inline Ws carve(float* ws, int C) {
    Ws w;
    w.partial = ws;
    w.sums = w.partial + (size_t)MAX_BLOCKS * 2 * C;
    w.scale = w.sums + 2 * C;
    w.shift = w.scale + C;
    w.mg = w.shift + C;
    w.mgx = w.mg + C;
    return w;
}

// column sums of the [blocks][2C] partials, 16 columns x 64 row lanes per block (fixed order)
inline void sum_partials(const Ws& w, int blocks, int C, hipStream_t st) { launch_colsum(w.partial, blocks, 2 * C, w.sums, st); }

}  // namespace

extern "C" int pswin_bn_workspace(int C) { return C > 0 ? (MAX_BLOCKS * 2 + 6) * C : PSWIN_ERR_ARG; }

extern "C" int pswin_bn_relu_fwd(const void* y, int dtype, const float* gamma, const float* beta,
                                 const float* mean_offset, float eps, float momentum, int train, float* running_mean, float* running_var, void* z,
                                 float* save_mean, float* save_rstd, float* workspace, long long M, int C,
                                 void* stream) {
    PSWIN_CHECK_ARG(y && gamma && beta && z && save_mean && save_rstd && workspace);
    PSWIN_CHECK_ARG(bn_args_ok(M, C, dtype) && aligned16(y) && aligned16(z));
    PSWIN_CHECK_ARG(train || (running_mean && running_var));
    hipStream_t st = (hipStream_t)stream;
    Ws w = carve(workspace, C);
    if (train) {
        const int blocks = reduce_blocks(M, C, dtype);
        if (dtype == PSWIN_BF16)
            hipLaunchKernelGGL((bn_reduce_kernel<PSWIN_BF16, 0>), dim3(blocks), dim3(THREADS), 0, st, y, nullptr, nullptr,
                               nullptr, nullptr, nullptr, M, C, w.partial);
        else
            hipLaunchKernelGGL((bn_reduce_kernel<PSWIN_F32, 0>), dim3(blocks), dim3(THREADS), 0, st, y, nullptr, nullptr,
                               nullptr, nullptr, nullptr, M, C, w.partial);
        sum_partials(w, blocks, C, st);
        hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, st, w.sums, M, C, gamma, beta,
                           mean_offset, eps, momentum, running_mean, running_var, save_mean, save_rstd, w.scale, w.shift);
    } else {
        hipLaunchKernelGGL(bn_eval_params_kernel, dim3((C + 63) / 64), dim3(64), 0, st, C, gamma, beta, mean_offset, eps,
                           running_mean, running_var, save_mean, save_rstd, w.scale, w.shift);
    }
    const int ve = dtype == PSWIN_BF16 ? 8 : 4;
    const long long n_vec = M * (C / ve);
    if (dtype == PSWIN_BF16)
        hipLaunchKernelGGL(bn_relu_apply_kernel<PSWIN_BF16>, dim3(apply_blocks(n_vec)), dim3(THREADS), 0, st, y, w.scale,
                           w.shift, z, n_vec, C);
    else
        hipLaunchKernelGGL(bn_relu_apply_kernel<PSWIN_F32>, dim3(apply_blocks(n_vec)), dim3(THREADS), 0, st, y, w.scale,
                           w.shift, z, n_vec, C);
    PSWIN_LAUNCH_RET();
}

extern "C" int pswin_bn_relu_bwd(const void* dz, const void* y, int dtype, const float* gamma, const float* beta,
                                 const float* save_mean, const float* save_rstd, int train, void* dy, float* dgamma,
                                 float* dbeta, float* workspace, long long M, int C, void* stream) {
    PSWIN_CHECK_ARG(dz && y && gamma && beta && save_mean && save_rstd && dy && dgamma && dbeta && workspace);
    PSWIN_CHECK_ARG(bn_args_ok(M, C, dtype) && aligned16(y) && aligned16(dz) && aligned16(dy));
    hipStream_t st = (hipStream_t)stream;
    Ws w = carve(workspace, C);
    // scale / shift are recomputed from the saved statistics (the forward workspace may have been reused since)
    hipLaunchKernelGGL(bn_scale_shift_kernel, dim3((C + 63) / 64), dim3(64), 0, st, C, gamma, beta, save_mean, save_rstd,
                       w.scale, w.shift);
    const int blocks = reduce_blocks(M, C, dtype);
    if (dtype == PSWIN_BF16)
        hipLaunchKernelGGL((bn_reduce_kernel<PSWIN_BF16, 1>), dim3(blocks), dim3(THREADS), 0, st, y, dz, w.scale, w.shift,
                           save_mean, save_rstd, M, C, w.partial);
    else
        hipLaunchKernelGGL((bn_reduce_kernel<PSWIN_F32, 1>), dim3(blocks), dim3(THREADS), 0, st, y, dz, w.scale, w.shift,
                           save_mean, save_rstd, M, C, w.partial);
    sum_partials(w, blocks, C, st);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, st, w.sums, M, C, train, dgamma, dbeta,
                       w.mg, w.mgx);
    const int ve = dtype == PSWIN_BF16 ? 8 : 4;
    const long long n_vec = M * (C / ve);
    if (dtype == PSWIN_BF16)
        hipLaunchKernelGGL(bn_relu_bwd_apply_kernel<PSWIN_BF16>, dim3(apply_blocks(n_vec)), dim3(THREADS), 0, st, dz, y,
                           w.scale, w.shift, save_mean, save_rstd, w.mg, w.mgx, dy, n_vec, C);
    else
        hipLaunchKernelGGL(bn_relu_bwd_apply_kernel<PSWIN_F32>, dim3(apply_blocks(n_vec)), dim3(THREADS), 0, st, dz, y,
                           w.scale, w.shift, save_mean, save_rstd, w.mg, w.mgx, dy, n_vec, C);
    PSWIN_LAUNCH_RET();
}
