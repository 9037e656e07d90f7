// AdamW over the ONE flat fp32 parameter buffer of a model (dp.GradReducer.flatten_parameters), gfx950.
//
// A training step of the hot path ends in the optimizer update (bench.py's step = fwd + bwd + AdamW, as the reference's
// configs/swin/*.py run it, mmdet/apis/train.py:91-112) and begins with the bf16 copies of the Linear weights that the bf16 kernels
// read.  Over one flat buffer both are a single streaming pass: p, g, m, v in; p, m, v and bf16(p) out -- 30 bytes per parameter,
// one launch, instead of the framework's two multi-tensor launches (28 B) plus a cast pass (6 B) at the start of the next step.
// Arithmetic = torch.optim.AdamW (decoupled weight decay, bias-corrected moments), f32, in this order:
//   p -= lr * wd * p;   m += (1 - b1) * (g - m);   v = b2 * v + (1 - b2) * g * g;
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// The step number t lives in device memory (a captured hipGraph replays the launch with the same arguments).
#include "pswin_common.hpp"

using namespace pswin;

namespace {

constexpr int MAX_GROUPS = PSWIN_ADAMW_MAX_GROUPS;
struct GroupMults {
    float lr[MAX_GROUPS], decay[MAX_GROUPS];       // per parameter group: multipliers of the base lr / weight decay
};

// GROUPS: group_of[i] names the parameter group of elements 4 i .. 4 i + 3 (one byte per 16-byte granule: 0.8 % more traffic);
// the reference's paramwise_cfg (decay_mult = 0 for norm layers / position tables) is two groups
template <bool GROUPS>
__global__ __launch_bounds__(256) void adamw_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, unsigned short* __restrict__ lowp, long long n4, double lr,
                                                         double b1d, double b2d, float eps, double wd, const float* __restrict__ step,
                                                         const unsigned char* __restrict__ group_of, const GroupMults gm) {
    // the hyper-parameters arrive as doubles (as torch hands them to its kernel) and every derived constant is formed in double
    // before it is rounded to f32: 1 - 0.999f is 1.3e-5 away from 1 - 0.999
    const double t = (double)*step;
    const double bc1 = 1.0 - pow(b1d, t), bc2 = 1.0 - pow(b2d, t);
    const float bc2s = (float)sqrt(bc2), w1 = (float)(1.0 - b1d), w2 = (float)(1.0 - b2d), b2 = (float)b2d;
    float step_size = (float)(lr / bc1), decay = (float)(lr * wd);
    [[maybe_unused]] float step_g[MAX_GROUPS], decay_g[MAX_GROUPS];
    if constexpr (GROUPS) {
#pragma unroll
        for (int k = 0; k < MAX_GROUPS; ++k) {        // per group exactly what torch computes for a group with lr * lr_mult, wd * decay_mult
            const double lrk = lr * (double)gm.lr[k];
            step_g[k] = (float)(lrk / bc1);
            decay_g[k] = (float)(lrk * (wd * (double)gm.decay[k]));
        }
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 pv = reinterpret_cast<const f32x4*>(p)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 mv = reinterpret_cast<const f32x4*>(m)[i], vv = reinterpret_cast<const f32x4*>(v)[i];
        if constexpr (GROUPS) {
            const int k = group_of[i];
            step_size = step_g[0];
            decay = decay_g[0];
#pragma unroll
            for (int q = 1; q < MAX_GROUPS; ++q) {    // select, not index: keeps the tables in registers
                step_size = k == q ? step_g[q] : step_size;
                decay = k == q ? decay_g[q] : decay;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pe = pv[e];
            pe -= decay * pe;
            const float me = mv[e] + w1 * (gv[e] - mv[e]);
            const float ve = b2 * vv[e] + w2 * gv[e] * gv[e];
            const float denom = sqrtf(ve) / bc2s + eps;
            pe -= step_size * me / denom;
            pv[e] = pe;
            mv[e] = me;
            vv[e] = ve;
        }
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(v)[i] = vv;
        if (lowp) {
            u32x2 o = {pack2_bf16(pv[0], pv[1]), pack2_bf16(pv[2], pv[3])};
            reinterpret_cast<u32x2*>(lowp)[i] = o;
        }
    }
}

}  // namespace

extern "C" int pswin_adamw_flat_groups(float* p, const float* g, float* m, float* v, void* p_bf16, long long n, const unsigned char* group_of,
                                       int n_groups, const float* lr_mult, const float* decay_mult, double lr, double beta1, double beta2,
                                       double eps, double weight_decay, const float* step, void* stream) {
    PSWIN_CHECK_ARG(p && g && m && v && step && n > 0 && n % 4 == 0);
    PSWIN_CHECK_ARG(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v) && (reinterpret_cast<uintptr_t>(p_bf16) & 7) == 0);
    PSWIN_CHECK_ARG(lr >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0.);
    PSWIN_CHECK_ARG(group_of ? (n_groups >= 1 && n_groups <= MAX_GROUPS && lr_mult && decay_mult) : n_groups == 0);
    const long long n4 = n / 4;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;             // 16 workgroups of 4 waves per CU, grid-stride over the rest
    GroupMults gm;
    for (int k = 0; k < MAX_GROUPS; ++k) {
        gm.lr[k] = (group_of && k < n_groups) ? lr_mult[k] : 1.f;
        gm.decay[k] = (group_of && k < n_groups) ? decay_mult[k] : 1.f;
        PSWIN_CHECK_ARG(gm.lr[k] >= 0.f && gm.decay[k] >= 0.f);
    }
    if (group_of)
        hipLaunchKernelGGL(adamw_flat_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                           reinterpret_cast<unsigned short*>(p_bf16), n4, lr, beta1, beta2, (float)eps, weight_decay, step, group_of, gm);
    else
        hipLaunchKernelGGL(adamw_flat_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                           reinterpret_cast<unsigned short*>(p_bf16), n4, lr, beta1, beta2, (float)eps, weight_decay, step, group_of, gm);
    PSWIN_LAUNCH_RET();
}

extern "C" int pswin_adamw_flat(float* p, const float* g, float* m, float* v, void* p_bf16, long long n, double lr, double beta1, double beta2,
                                double eps, double weight_decay, const float* step, void* stream) {
    return pswin_adamw_flat_groups(p, g, m, v, p_bf16, n, nullptr, 0, nullptr, nullptr, lr, beta1, beta2, eps, weight_decay, step, stream);
}
