// Shared device/host helpers for libpswin_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "pswin.h"

#define PSWIN_CHECK_ARG(cond) \
    do {                      \
        if (!(cond)) return PSWIN_ERR_ARG; \
    } while (0)

#define PSWIN_LAUNCH_RET()                              \
    do {                                                \
        hipError_t e__ = hipGetLastError();             \
        return e__ == hipSuccess ? PSWIN_OK : (int)e__; \
    } while (0)

namespace pswin {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__host__ __device__ inline int ceil_to(int v, int m) { return (v + m - 1) / m * m; }

__device__ inline float bf16_bits_to_f32(unsigned short b) {
    return __builtin_bit_cast(float, (unsigned int)b << 16);
}
// round-to-nearest-even f32 -> bf16 (hipcc lowers the cast to v_cvt_pk_bf16_f32 on gfx950, NaN-preserving)
__device__ inline unsigned short f32_to_bf16_bits(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
// two f32 -> one dword of two bf16 (lo in bits 0-15) with ONE v_cvt_pk_bf16_f32: `f32_to_bf16_bits(lo) | f32_to_bf16_bits(hi) << 16`
// compiles to two conversions, a shift and an or (seen in the ISA of every kernel that packed its outputs that way)
typedef __attribute__((ext_vector_type(2))) float pswin_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 pswin_bf16x2_t;
__device__ inline unsigned int pack2_bf16(float lo, float hi) {
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(pswin_f32x2_t{lo, hi}, pswin_bf16x2_t));
}

// 4 consecutive elements of a row, as f32, from an f32 or bf16 buffer
template <int DT>
__device__ inline f32x4 load4(const void* base, size_t elem_off) {
    if constexpr (DT == PSWIN_F32) {
        return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + elem_off);
    } else {
        u32x2 raw = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(base) + elem_off);
        f32x4 r;
        r[0] = __builtin_bit_cast(float, raw[0] << 16);
        r[1] = __builtin_bit_cast(float, raw[0] & 0xffff0000u);
        r[2] = __builtin_bit_cast(float, raw[1] << 16);
        r[3] = __builtin_bit_cast(float, raw[1] & 0xffff0000u);
        return r;
    }
}

template <int DT>
__device__ inline void store4(void* base, size_t elem_off, f32x4 v) {
    if constexpr (DT == PSWIN_F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + elem_off) = v;
    } else {
        u32x2 raw;
        raw[0] = pack2_bf16(v[0], v[1]);
        raw[1] = pack2_bf16(v[2], v[3]);
        *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(base) + elem_off) = raw;
    }
}

// out[c] = sum_r part[r][c] (r in fixed order: bitwise reproducible).  16 columns x 64 row lanes per block, so that
// even 512 partial rows are only 8 (independent) loads deep per thread: this stage is pure latency.
__global__ static void colsum_kernel(const float* __restrict__ part, int R, int N, float* __restrict__ out) {
    __shared__ float red[64][16];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < N) {
        int r = rl;
        for (; r + 192 < R; r += 256) {
            s0 += part[(size_t)r * N + c];
            s1 += part[(size_t)(r + 64) * N + c];
            s2 += part[(size_t)(r + 128) * N + c];
            s3 += part[(size_t)(r + 192) * N + c];
        }
        for (; r < R; r += 64) s0 += part[(size_t)r * N + c];
    }
    red[rl][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl < 16) {      // 16 x 16 threads finish: thread (rl, cl) sums rows rl, rl+16, rl+32, rl+48, then a 16-lane tree
        float s = (red[rl][cl] + red[rl + 16][cl]) + (red[rl + 32][cl] + red[rl + 48][cl]);
        red[rl][cl] = s;
    }
    __syncthreads();
    if (rl == 0 && c < N) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][cl];
        out[c] = s;
    }
}

inline void launch_colsum(const float* part, int R, int N, float* out, hipStream_t st) {
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 15) / 16), dim3(1024), 0, st, part, R, N, out);
}

// Opt a kernel in to `bytes` of dynamic LDS (> 64 KB needs hipFuncAttributeMaxDynamicSharedMemorySize) on the CURRENT device.
// The attribute is per device, so the "already done" memo is a bit per device ordinal (lock-free; setting it twice is harmless);
// a failure is returned to the caller instead of surfacing later as an unexplained launch error.
inline int ensure_dynamic_lds(const void* kernel, size_t bytes, std::atomic<unsigned long long>& done) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const bool memo = dev >= 0 && dev < 64;
    const unsigned long long bit = memo ? 1ull << dev : 0ull;
    if (memo && (done.load(std::memory_order_acquire) & bit)) return PSWIN_OK;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    if (memo) done.fetch_or(bit, std::memory_order_release);
    return PSWIN_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool valid_dtype(int dt) { return dt == PSWIN_F32 || dt == PSWIN_BF16; }

}  // namespace pswin
