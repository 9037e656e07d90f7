// Spherical geometry of the PanoSwin hot path (fp32): uv grid, absolute-position features, great-circle
// (haversine) distances per window.  All of it is input independent, so the host caches the results per
// feature-map shape; none of these kernels sits on the per-step critical path.
//
// Reference: make_uv_hw2 (HOT:153-189), _pano_abs_position (HOT:909-938), haversine22
// (lzx/models/great_circle.py:71-86).  HOT = mmdet/models/backbones/simple_panoswin_transformer.py.
#include "pswin_common.hpp"

using namespace pswin;

namespace {

// Same rounding sequence as the reference's fp32 tensor ops: (float(x) * gap - c) + gap/2.
// __fmul_rn / __fsub_rn / __fadd_rn are never contracted into FMAs.
__global__ void uv_grid_kernel(int H, int W, float gap, float half_gap, float pi_f, float half_pi_f,
                               float* __restrict__ uv) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= H * W) return;
    int y = t / W, x = t - y * W;
    float u = __fadd_rn(__fsub_rn(__fmul_rn((float)x, gap), pi_f), half_gap);
    float v = __fadd_rn(__fsub_rn(__fmul_rn((float)y, gap), half_pi_f), half_gap);
    uv[2 * t] = u;
    uv[2 * t + 1] = v;
}

__global__ void abs_pos_kernel(const float* __restrict__ uv, int n, float* __restrict__ feat) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float u = uv[2 * t], v = uv[2 * t + 1];
    float su = sinf(u), cu = cosf(u), sv = sinf(v), cv = cosf(v);
    feat[5 * t + 0] = __fmul_rn(su, sv);
    feat[5 * t + 1] = __fmul_rn(cu, sv);
    feat[5 * t + 2] = cv;
    feat[5 * t + 3] = u;
    feat[5 * t + 4] = v;
}

__global__ void gather_uv_kernel(const float* __restrict__ uv, const int32_t* __restrict__ map, int n_slots,
                                 float* __restrict__ uv_win) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    int src = map[s];
    f32x2 val = {0.f, 0.f};
    if (src >= 0) val = *reinterpret_cast<const f32x2*>(uv + 2 * (size_t)src);
    *reinterpret_cast<f32x2*>(uv_win + 2 * (size_t)s) = val;
}

// one block per window; uv of both sides staged in LDS with the per-token cosines precomputed
__global__ void haversine_kernel(const float* __restrict__ uv1, const float* __restrict__ uv2,
                                 float* __restrict__ dist) {
    __shared__ float u1[PSWIN_WTOK], v1[PSWIN_WTOK], c1[PSWIN_WTOK];
    __shared__ float u2[PSWIN_WTOK], v2[PSWIN_WTOK], c2[PSWIN_WTOK];
    size_t w = blockIdx.x;
    for (int t = threadIdx.x; t < PSWIN_WTOK; t += blockDim.x) {
        float a = uv1[(w * PSWIN_WTOK + t) * 2], b = uv1[(w * PSWIN_WTOK + t) * 2 + 1];
        u1[t] = a; v1[t] = b; c1[t] = cosf(b);
        a = uv2[(w * PSWIN_WTOK + t) * 2]; b = uv2[(w * PSWIN_WTOK + t) * 2 + 1];
        u2[t] = a; v2[t] = b; c2[t] = cosf(b);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < PSWIN_WTOK * PSWIN_WTOK; e += blockDim.x) {
        int i = e / PSWIN_WTOK, j = e - i * PSWIN_WTOK;
        float sdv = sinf(__fmul_rn(0.5f, fabsf(__fsub_rn(v2[j], v1[i]))));
        float sdu = sinf(__fmul_rn(0.5f, __fsub_rn(u2[j], u1[i])));
        float a = __fadd_rn(__fmul_rn(sdv, sdv), __fmul_rn(__fmul_rn(c2[j], c1[i]), __fmul_rn(sdu, sdu)));
        dist[w * PSWIN_WTOK * PSWIN_WTOK + e] = __fmul_rn(asinf(sqrtf(a)), 2.0f);
    }
}

}  // namespace

extern "C" int pswin_uv_grid(int H, int W, float* uv, void* stream) {
    PSWIN_CHECK_ARG(H > 0 && W > 0 && uv != nullptr);
    PSWIN_CHECK_ARG(H <= W);  // make_uv_hw2 slices arange(W)[:H] (HOT:174-175)
    const double pi = 3.14159265358979323846;
    double gap = pi / H;
    int n = H * W;
    hipLaunchKernelGGL(uv_grid_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, H, W, (float)gap,
                       (float)(0.5 * gap), (float)pi, (float)(pi * 0.5), uv);
    PSWIN_LAUNCH_RET();
}

extern "C" int pswin_abs_pos_features(const float* uv, int n, float* feat, void* stream) {
    PSWIN_CHECK_ARG(uv != nullptr && feat != nullptr && n > 0);
    hipLaunchKernelGGL(abs_pos_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, uv, n, feat);
    PSWIN_LAUNCH_RET();
}

extern "C" int pswin_gather_uv(const float* uv, const int32_t* map, int n_slots, float* uv_win, void* stream) {
    PSWIN_CHECK_ARG(uv != nullptr && map != nullptr && uv_win != nullptr && n_slots > 0);
    hipLaunchKernelGGL(gather_uv_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, (hipStream_t)stream, uv, map,
                       n_slots, uv_win);
    PSWIN_LAUNCH_RET();
}

extern "C" int pswin_haversine_windows(const float* uv1, const float* uv2, int n_windows, float* dist,
                                       void* stream) {
    PSWIN_CHECK_ARG(uv1 != nullptr && uv2 != nullptr && dist != nullptr && n_windows > 0);
    hipLaunchKernelGGL(haversine_kernel, dim3(n_windows), dim3(256), 0, (hipStream_t)stream, uv1, uv2, dist);
    PSWIN_LAUNCH_RET();
}
