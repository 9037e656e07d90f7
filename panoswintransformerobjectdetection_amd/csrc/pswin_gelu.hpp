// Exact GELU (erf form) and its derivative for the bf16 kernels (gfx950): shared by pswin_mlp.hip and pswin_gemm.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace pswin {

// Phi(v) = 0.5 (1 + erf(v / sqrt 2)) and E = exp(-v^2 / 2).  The GELU kernels are VALU bound with libm's erff / expf
// (~50 instructions per element: 100 M elements of a stage-0 block take 150 us of pure ALU time, the memory traffic
// 100 us), so erf uses Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. f32 rounding level) on the hardware
// reciprocal and exp2, sharing ONE exponential with the density term of the derivative:
//   erf(z) = 1 - (a1 t + a2 t^2 + a3 t^3 + a4 t^4 + a5 t^5) exp(-z^2),  t = 1 / (1 + p z),  z = |v| / sqrt 2 >= 0
// and the lower tail is formed directly (0.5 poly E, no 1 - erf cancellation).
__device__ inline void phi_and_exp(float v, float& cdf, float& E) {
    const float z = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    E = __builtin_amdgcn_exp2f(v * v * -0.72134752044448170368f);            // exp(-v^2 / 2)
    float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    const float tail = 0.5f * p * t * E;                                        // 0.5 erfc(z)
    cdf = v >= 0.f ? 1.0f - tail : tail;
}
__device__ inline float gelu_f(float v) {
    float cdf, E;
    phi_and_exp(v, cdf, E);
    return v * cdf;
}
__device__ inline float gelu_grad_f(float v) {
    float cdf, E;
    phi_and_exp(v, cdf, E);
    return __builtin_fmaf(v * 0.39894228040143267794f, E, cdf);
}

// Two elements at a time on packed-f32 VALU instructions (v_pk_mul_f32 / v_pk_fma_f32): the polynomial and the scalings
// take half the issue slots; the reciprocal and the exponential stay scalar (transcendental unit).  Used by the kernels
// whose epilogue is VALU bound (the fused fc1 + GELU streaming GEMM).
typedef __attribute__((ext_vector_type(2))) float gelu_f32x2;
__device__ inline void phi_and_exp2(gelu_f32x2 v, gelu_f32x2& cdf, gelu_f32x2& E) {
    const gelu_f32x2 az = {fabsf(v[0]), fabsf(v[1])};
    const gelu_f32x2 den = __builtin_elementwise_fma(az, gelu_f32x2{0.3275911f * 0.70710678118654752440f, 0.3275911f * 0.70710678118654752440f},
                                                     gelu_f32x2{1.0f, 1.0f});
    const gelu_f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    const gelu_f32x2 ex = v * v * gelu_f32x2{-0.72134752044448170368f, -0.72134752044448170368f};
    E = gelu_f32x2{__builtin_amdgcn_exp2f(ex[0]), __builtin_amdgcn_exp2f(ex[1])};
    gelu_f32x2 p = __builtin_elementwise_fma(gelu_f32x2{1.061405429f, 1.061405429f}, t, gelu_f32x2{-1.453152027f, -1.453152027f});
    p = __builtin_elementwise_fma(p, t, gelu_f32x2{1.421413741f, 1.421413741f});
    p = __builtin_elementwise_fma(p, t, gelu_f32x2{-0.284496736f, -0.284496736f});
    p = __builtin_elementwise_fma(p, t, gelu_f32x2{0.254829592f, 0.254829592f});
    const gelu_f32x2 tail = p * t * E * gelu_f32x2{0.5f, 0.5f};
    cdf = gelu_f32x2{v[0] >= 0.f ? 1.0f - tail[0] : tail[0], v[1] >= 0.f ? 1.0f - tail[1] : tail[1]};
}
__device__ inline gelu_f32x2 gelu_f2(gelu_f32x2 v) {
    gelu_f32x2 cdf, E;
    phi_and_exp2(v, cdf, E);
    return v * cdf;
}
__device__ inline gelu_f32x2 gelu_grad_f2(gelu_f32x2 v) {
    gelu_f32x2 cdf, E;
    phi_and_exp2(v, cdf, E);
    return __builtin_elementwise_fma(v * gelu_f32x2{0.39894228040143267794f, 0.39894228040143267794f}, E, cdf);
}

}  // namespace pswin
