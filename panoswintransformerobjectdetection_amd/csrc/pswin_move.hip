// Row movers: every data-layout step of a PanoSwin block as ONE indexed row copy.
//
// The reference materialises, per block, roll -> flip -> cat -> roll -> pad -> view/permute/contiguous on
// the way in (WindowTransition.forward, pad_x, window_partition: HOT:393-406, 486-491, 64-75) and the
// inverse chain plus the residual add on the way out (HOT:78-92, 516-533): ~12 full-tensor copies.  Here a
// precomputed int32 window map (pswin_index.hip) turns each direction into a single HBM-bound gather of
// C-element rows, fused with the dtype conversion, the DropPath per-sample scale and the residual add.
// PatchMerging's strided 2x2 gather + concat (HOT:563-572) and the two static F.grid_sample calls of the
// pitch module (HOT:1038, 1090) are the same kind of kernel.
// HOT = mmdet/models/backbones/simple_panoswin_transformer.py of the reference.
//
// Thread mapping: a row of C elements is C/4 lanes x 4 elements (16 B f32 / 8 B bf16 per lane, consecutive
// lanes on consecutive addresses); a 256-thread block covers 256/(C/4) rows.
#include "pswin_common.hpp"
#include <type_traits>

using namespace pswin;

namespace {

struct RowGeom {
    int vpr;   // 4-element vectors per row
    dim3 block;
    unsigned rows_per_block;
};

inline RowGeom row_geom(int C) {
    RowGeom g;
    g.vpr = C / 4;
    unsigned bx = g.vpr < 256 ? g.vpr : 256;
    unsigned by = 256 / bx;
    if (by < 1) by = 1;
    g.block = dim3(bx, by);
    g.rows_per_block = by;
    return g;
}

template <int XDT, int WDT>
__global__ void window_gather_kernel(const void* __restrict__ x, const int32_t* __restrict__ map,
                                     const float* __restrict__ scale, void* __restrict__ win, long long rows,
                                     int S, int n_slots, int vpr) {
    long long row = (long long)blockIdx.x * blockDim.y + threadIdx.y;
    if (row >= rows) return;
    int b = (int)(row / n_slots);
    int slot = (int)(row - (long long)b * n_slots);
    int src = map[slot];
    float sc = scale ? scale[b] : 1.0f;
    size_t C = (size_t)vpr * 4;
    for (int v = threadIdx.x; v < vpr; v += blockDim.x) {
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (src >= 0) {
            val = load4<XDT>(x, ((size_t)b * S + src) * C + 4 * (size_t)v);
            if (scale) val = val * sc;
        }
        store4<WDT>(win, (size_t)row * C + 4 * (size_t)v, val);
    }
}

// The two movers of the bf16 training step, 8 elements per thread and two rows per thread: one 16-byte access on the
// bf16 side (the generic kernels above move 8 bytes per thread there), two on the f32 side, two independent rows in
// flight.  Same arithmetic, element for element, as the generic kernels.
//   gather8:  win[b][slot] = bf16(scale_b * x[b][map[slot]])          x: f32, win: bf16   (backward of scatter_add)
//   scatter8: out[b][t] = resid[b][t] + scale_b * (win[b][inv[t]] + bias)   win: bf16, resid / out: f32
// Both kernels handle two rows per thread in PHASES: the map entries and everything that does not depend on them (scales,
// residual rows) are requested for both rows at once, then the mapped rows, then the arithmetic and the stores.  Written row
// after row, each row was a chain of up to four dependent memory round trips and the second row's loads queued behind the
// first row's stores (one in-order counter).
// Straight-line code (template flags instead of pointer tests, rows past the end clamped instead of skipped): a branch around a
// load is a join where the compiler waits for everything outstanding.
template <bool SCALE>
__global__ __launch_bounds__(256) void window_gather8_kernel(const float* __restrict__ x, const int32_t* __restrict__ map,
                                                             const float* __restrict__ scale,
                                                             unsigned short* __restrict__ win, long long rows, int S,
                                                             int n_slots, int lanes, int rpb) {
    const int lane = threadIdx.x % lanes, rl = threadIdx.x / lanes;
    if (rl >= rpb) return;
    const size_t C = (size_t)lanes * 8;
    // blockIdx.y = image: no division, and a row index past the image's last slot is clamped for the loads and skips the store
    const int b = blockIdx.y;
    unsigned r[2];
    int src[2];
    const float sc = SCALE ? scale[b] : 1.0f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        r[u] = (blockIdx.x * 2u + u) * (unsigned)rpb + (unsigned)rl;
        src[u] = map[r[u] < (unsigned)n_slots ? r[u] : (unsigned)n_slots - 1];
    }
    asm volatile("" : "+v"(src[0]), "+v"(src[1]));          // both map entries requested before either is used (the compiler would
                                                            // sink a row's loads into the branch around its store)
    f32x4 a[2], c[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const float* p = x + ((size_t)b * S + (src[u] >= 0 ? src[u] : 0)) * C + 8 * (size_t)lane;         // padding slots: any valid row
        a[u] = *reinterpret_cast<const f32x4*>(p);
        c[u] = *reinterpret_cast<const f32x4*>(p + 4);
    }
    asm volatile("" : "+v"(a[0]), "+v"(c[0]), "+v"(a[1]), "+v"(c[1]));
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const float m = src[u] >= 0 ? sc : 0.f;                                                            // padding slots are zero rows
        const f32x4 av = (SCALE || src[u] < 0) ? a[u] * m : a[u], cv = (SCALE || src[u] < 0) ? c[u] * m : c[u];
        u32x4 o;
        o[0] = pack2_bf16(av[0], av[1]);
        o[1] = pack2_bf16(av[2], av[3]);
        o[2] = pack2_bf16(cv[0], cv[1]);
        o[3] = pack2_bf16(cv[2], cv[3]);
        if (src[u] < 0) o = u32x4{0u, 0u, 0u, 0u};                                                       // exact zeros (x may hold inf / nan)
        if (r[u] < (unsigned)n_slots) *reinterpret_cast<u32x4*>(win + ((size_t)b * n_slots + r[u]) * C + 8 * (size_t)lane) = o;
    }
}

template <bool SCALE, bool RESID, bool BIAS>
__global__ __launch_bounds__(256) void window_scatter_add8_kernel(const unsigned short* __restrict__ win,
                                                                  const int32_t* __restrict__ inv,
                                                                  const float* __restrict__ resid,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ bias, float* __restrict__ out,
                                                                  long long rows, int S, int n_slots, int lanes, int rpb) {
    const int lane = threadIdx.x % lanes, rl = threadIdx.x / lanes;
    if (rl >= rpb) return;
    const size_t C = (size_t)lanes * 8;
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if constexpr (BIAS) {
        b0 = *reinterpret_cast<const f32x4*>(bias + 8 * lane);
        b1 = *reinterpret_cast<const f32x4*>(bias + 8 * lane + 4);
    }
    // blockIdx.y = image: no division, and a token index past the image's last token is clamped for the loads and skips the store
    const int b = blockIdx.y;
    unsigned t[2], tc[2];
    int slot[2];
    const float sc = SCALE ? scale[b] : 1.0f;
    f32x4 r0[2], r1[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        t[u] = (blockIdx.x * 2u + u) * (unsigned)rpb + (unsigned)rl;
        tc[u] = t[u] < (unsigned)S ? t[u] : (unsigned)S - 1;
        slot[u] = inv[tc[u]];
        r0[u] = r1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (RESID) {
            const size_t o = ((size_t)b * S + tc[u]) * C + 8 * (size_t)lane;
            r0[u] = *reinterpret_cast<const f32x4*>(resid + o);
            r1[u] = *reinterpret_cast<const f32x4*>(resid + o + 4);
        }
    }
    asm volatile("" : "+v"(slot[0]), "+v"(slot[1]));        // both map entries (and the residual rows) requested before either is used
    u32x4 raw[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) raw[u] = *reinterpret_cast<const u32x4*>(win + ((size_t)b * n_slots + slot[u]) * C + 8 * (size_t)lane);
    asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(r0[0]), "+v"(r1[0]), "+v"(r0[1]), "+v"(r1[1]));
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        f32x4 a = {__builtin_bit_cast(float, raw[u][0] << 16), __builtin_bit_cast(float, raw[u][0] & 0xffff0000u),
                   __builtin_bit_cast(float, raw[u][1] << 16), __builtin_bit_cast(float, raw[u][1] & 0xffff0000u)};
        f32x4 c = {__builtin_bit_cast(float, raw[u][2] << 16), __builtin_bit_cast(float, raw[u][2] & 0xffff0000u),
                   __builtin_bit_cast(float, raw[u][3] << 16), __builtin_bit_cast(float, raw[u][3] & 0xffff0000u)};
        if constexpr (BIAS) {
            a = a + b0;
            c = c + b1;
        }
        if constexpr (SCALE) {
            a = a * sc;
            c = c * sc;
        }
        if constexpr (RESID) {
            a = a + r0[u];
            c = c + r1[u];
        }
        if (t[u] < (unsigned)S) {
            const size_t o = ((size_t)b * S + t[u]) * C + 8 * (size_t)lane;
            *reinterpret_cast<f32x4*>(out + o) = a;
            *reinterpret_cast<f32x4*>(out + o + 4) = c;
        }
    }
}

template <int WDT, int XDT>
__global__ void window_scatter_add_kernel(const void* __restrict__ win, const int32_t* __restrict__ inv,
                                          const void* __restrict__ resid, const float* __restrict__ scale,
                                          const float* __restrict__ bias, void* __restrict__ out, long long rows, int S,
                                          int n_slots, int vpr) {
    long long row = (long long)blockIdx.x * blockDim.y + threadIdx.y;
    if (row >= rows) return;
    int b = (int)(row / S);
    int t = (int)(row - (long long)b * S);
    int slot = inv[t];
    float sc = scale ? scale[b] : 1.0f;
    size_t C = (size_t)vpr * 4;
    for (int v = threadIdx.x; v < vpr; v += blockDim.x) {
        f32x4 val = load4<WDT>(win, ((size_t)b * n_slots + slot) * C + 4 * (size_t)v);
        if (bias) val = val + *reinterpret_cast<const f32x4*>(bias + 4 * (size_t)v);
        if (scale) val = val * sc;
        if (resid) val = val + load4<XDT>(resid, (size_t)row * C + 4 * (size_t)v);
        store4<XDT>(out, (size_t)row * C + 4 * (size_t)v, val);
    }
}

template <int XDT, int ODT>
__global__ void patch_merge_gather_kernel(const void* __restrict__ x, void* __restrict__ out, long long rows, int H,
                                          int W, int H2, int W2, int vpr) {
    // one "row" = one C-wide quarter of a merged token: rows = B * H2 * W2 * 4
    long long row = (long long)blockIdx.x * blockDim.y + threadIdx.y;
    if (row >= rows) return;
    int k = (int)(row & 3);
    long long tok = row >> 2;
    int b = (int)(tok / ((long long)H2 * W2));
    int r = (int)(tok - (long long)b * H2 * W2);
    int i = r / W2, j = r - i * W2;
    int yy = 2 * i + (k & 1), xx = 2 * j + (k >> 1);   // blocks: (0,0), (1,0), (0,1), (1,1) = (dy, dx)
    bool ok = yy < H && xx < W;
    size_t C = (size_t)vpr * 4;
    for (int v = threadIdx.x; v < vpr; v += blockDim.x) {
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (ok) val = load4<XDT>(x, ((size_t)b * H * W + (size_t)yy * W + xx) * C + 4 * (size_t)v);
        store4<ODT>(out, (size_t)row * C + 4 * (size_t)v, val);
    }
}

template <int ODT, int XDT>
__global__ void patch_merge_scatter_kernel(const void* __restrict__ dout, void* __restrict__ dx, long long rows,
                                           int H, int W, int H2, int W2, int vpr) {
    long long row = (long long)blockIdx.x * blockDim.y + threadIdx.y;   // rows = B * H * W
    if (row >= rows) return;
    int b = (int)(row / ((long long)H * W));
    int r = (int)(row - (long long)b * H * W);
    int h = r / W, w = r - h * W;
    int k = (h & 1) + 2 * (w & 1);
    size_t C = (size_t)vpr * 4;
    size_t src = (((size_t)b * H2 + (h >> 1)) * W2 + (w >> 1)) * 4 + k;
    for (int v = threadIdx.x; v < vpr; v += blockDim.x) {
        f32x4 val = load4<ODT>(dout, src * C + 4 * (size_t)v);
        store4<XDT>(dx, (size_t)row * C + 4 * (size_t)v, val);
    }
}

__global__ void interp_rows_kernel(const float* __restrict__ x, const int32_t* __restrict__ idx,
                                   const float* __restrict__ wgt, float* __restrict__ out, long long rows, int S,
                                   int P, int vpr) {
    long long row = (long long)blockIdx.x * blockDim.y + threadIdx.y;
    if (row >= rows) return;
    int b = (int)(row / P);
    int p = (int)(row - (long long)b * P);
    size_t C = (size_t)vpr * 4;
    int i0 = idx[4 * p], i1 = idx[4 * p + 1], i2 = idx[4 * p + 2], i3 = idx[4 * p + 3];
    float w0 = wgt[4 * p], w1 = wgt[4 * p + 1], w2 = wgt[4 * p + 2], w3 = wgt[4 * p + 3];
    const float* xb = x + (size_t)b * S * C;
    for (int v = threadIdx.x; v < vpr; v += blockDim.x) {
        // same accumulation order as ATen's bilinear grid_sample: nw, ne, sw, se
        f32x4 a = *reinterpret_cast<const f32x4*>(xb + (size_t)i0 * C + 4 * v) * w0;
        a = a + *reinterpret_cast<const f32x4*>(xb + (size_t)i1 * C + 4 * v) * w1;
        a = a + *reinterpret_cast<const f32x4*>(xb + (size_t)i2 * C + 4 * v) * w2;
        a = a + *reinterpret_cast<const f32x4*>(xb + (size_t)i3 * C + 4 * v) * w3;
        *reinterpret_cast<f32x4*>(out + (size_t)row * C + 4 * (size_t)v) = a;
    }
}

__global__ void interp_rows_adjoint_kernel(const float* __restrict__ dout, const int32_t* __restrict__ idx,
                                           const float* __restrict__ wgt, float* __restrict__ dx, long long rows,
                                           int S, int P, int vpr) {
    long long row = (long long)blockIdx.x * blockDim.y + threadIdx.y;
    if (row >= rows) return;
    int b = (int)(row / P);
    int p = (int)(row - (long long)b * P);
    size_t C = (size_t)vpr * 4;
    float* xb = dx + (size_t)b * S * C;
    for (int v = threadIdx.x; v < vpr; v += blockDim.x) {
        f32x4 g = *reinterpret_cast<const f32x4*>(dout + (size_t)row * C + 4 * (size_t)v);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float w = wgt[4 * p + k];
            if (w == 0.f) continue;
            float* dst = xb + (size_t)idx[4 * p + k] * C + 4 * v;
#pragma unroll
            for (int c = 0; c < 4; ++c) atomicAdd(dst + c, g[c] * w);
        }
    }
}

template <typename F>
inline int dispatch2(int a, int b, F&& f) {
    if (a == PSWIN_F32 && b == PSWIN_F32) return f(std::integral_constant<int, PSWIN_F32>(), std::integral_constant<int, PSWIN_F32>());
    if (a == PSWIN_F32 && b == PSWIN_BF16) return f(std::integral_constant<int, PSWIN_F32>(), std::integral_constant<int, PSWIN_BF16>());
    if (a == PSWIN_BF16 && b == PSWIN_F32) return f(std::integral_constant<int, PSWIN_BF16>(), std::integral_constant<int, PSWIN_F32>());
    return f(std::integral_constant<int, PSWIN_BF16>(), std::integral_constant<int, PSWIN_BF16>());
}

inline bool rows_ok(long long rows, unsigned rpb) { return rows > 0 && (rows + rpb - 1) / rpb < (1ll << 31); }

}  // namespace

extern "C" int pswin_window_gather(const void* x, int x_dtype, const int32_t* map, const float* scale, void* win,
                                   int win_dtype, int B, int S, int n_slots, int C, void* stream) {
    PSWIN_CHECK_ARG(x && map && win && B > 0 && S > 0 && n_slots > 0 && C > 0);
    PSWIN_CHECK_ARG(valid_dtype(x_dtype) && valid_dtype(win_dtype));
    PSWIN_CHECK_ARG(C % 8 == 0 && aligned16(x) && aligned16(win));
    RowGeom g = row_geom(C);
    long long rows = (long long)B * n_slots;
    PSWIN_CHECK_ARG(rows_ok(rows, g.rows_per_block));
    if (x_dtype == PSWIN_F32 && win_dtype == PSWIN_BF16 && C / 8 <= 256 && B <= 65535) {
        const int lanes = C / 8, rpb = 256 / lanes;
        const dim3 grid8((unsigned)((n_slots + 2 * rpb - 1) / (2 * rpb)), (unsigned)B);
        if (scale)
            hipLaunchKernelGGL(window_gather8_kernel<true>, grid8, dim3(256), 0, (hipStream_t)stream, (const float*)x, map, scale,
                               (unsigned short*)win, rows, S, n_slots, lanes, rpb);
        else
            hipLaunchKernelGGL(window_gather8_kernel<false>, grid8, dim3(256), 0, (hipStream_t)stream, (const float*)x, map, scale,
                               (unsigned short*)win, rows, S, n_slots, lanes, rpb);
        PSWIN_LAUNCH_RET();
    }
    dim3 grid((unsigned)((rows + g.rows_per_block - 1) / g.rows_per_block));
    return dispatch2(x_dtype, win_dtype, [&](auto xd, auto wd) {
        hipLaunchKernelGGL((window_gather_kernel<decltype(xd)::value, decltype(wd)::value>), grid, g.block, 0,
                           (hipStream_t)stream, x, map, scale, win, rows, S, n_slots, g.vpr);
        PSWIN_LAUNCH_RET();
    });
}

extern "C" int pswin_window_scatter_add(const void* win, int win_dtype, const int32_t* inv, const void* resid,
                                        const float* scale, const float* bias, void* out, int x_dtype, int B, int S,
                                        int n_slots, int C, void* stream) {
    PSWIN_CHECK_ARG(win && inv && out && B > 0 && S > 0 && n_slots > 0 && C > 0);
    PSWIN_CHECK_ARG(valid_dtype(x_dtype) && valid_dtype(win_dtype));
    PSWIN_CHECK_ARG(C % 8 == 0 && aligned16(win) && aligned16(out) && aligned16(resid) && aligned16(bias));
    RowGeom g = row_geom(C);
    long long rows = (long long)B * S;
    PSWIN_CHECK_ARG(rows_ok(rows, g.rows_per_block));
    if (win_dtype == PSWIN_BF16 && x_dtype == PSWIN_F32 && C / 8 <= 256 && B <= 65535) {
        const int lanes = C / 8, rpb = 256 / lanes;
        const dim3 grid8((unsigned)((S + 2 * rpb - 1) / (2 * rpb)), (unsigned)B);
        auto go = [&](auto sc_, auto rs_, auto bs_) {
            hipLaunchKernelGGL((window_scatter_add8_kernel<decltype(sc_)::value, decltype(rs_)::value, decltype(bs_)::value>), grid8, dim3(256),
                               0, (hipStream_t)stream, (const unsigned short*)win, inv, (const float*)resid, scale, bias, (float*)out,
                               rows, S, n_slots, lanes, rpb);
        };
        using T = std::true_type;
        using F = std::false_type;
        const int variant = (scale ? 4 : 0) | (resid ? 2 : 0) | (bias ? 1 : 0);
        switch (variant) {
            case 0: go(F{}, F{}, F{}); break;
            case 1: go(F{}, F{}, T{}); break;
            case 2: go(F{}, T{}, F{}); break;
            case 3: go(F{}, T{}, T{}); break;
            case 4: go(T{}, F{}, F{}); break;
            case 5: go(T{}, F{}, T{}); break;
            case 6: go(T{}, T{}, F{}); break;
            default: go(T{}, T{}, T{}); break;
        }
        PSWIN_LAUNCH_RET();
    }
    dim3 grid((unsigned)((rows + g.rows_per_block - 1) / g.rows_per_block));
    return dispatch2(win_dtype, x_dtype, [&](auto wd, auto xd) {
        hipLaunchKernelGGL((window_scatter_add_kernel<decltype(wd)::value, decltype(xd)::value>), grid, g.block, 0,
                           (hipStream_t)stream, win, inv, resid, scale, bias, out, rows, S, n_slots, g.vpr);
        PSWIN_LAUNCH_RET();
    });
}

extern "C" int pswin_patch_merge_gather(const void* x, int x_dtype, void* out, int out_dtype, int B, int H, int W,
                                        int C, void* stream) {
    PSWIN_CHECK_ARG(x && out && B > 0 && H > 0 && W > 0 && C > 0);
    PSWIN_CHECK_ARG(valid_dtype(x_dtype) && valid_dtype(out_dtype));
    PSWIN_CHECK_ARG(C % 8 == 0 && aligned16(x) && aligned16(out));
    RowGeom g = row_geom(C);
    int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
    long long rows = (long long)B * H2 * W2 * 4;
    PSWIN_CHECK_ARG(rows_ok(rows, g.rows_per_block));
    dim3 grid((unsigned)((rows + g.rows_per_block - 1) / g.rows_per_block));
    return dispatch2(x_dtype, out_dtype, [&](auto xd, auto od) {
        hipLaunchKernelGGL((patch_merge_gather_kernel<decltype(xd)::value, decltype(od)::value>), grid, g.block, 0,
                           (hipStream_t)stream, x, out, rows, H, W, H2, W2, g.vpr);
        PSWIN_LAUNCH_RET();
    });
}

extern "C" int pswin_patch_merge_scatter(const void* dout, int out_dtype, void* dx, int x_dtype, int B, int H, int W,
                                         int C, void* stream) {
    PSWIN_CHECK_ARG(dout && dx && B > 0 && H > 0 && W > 0 && C > 0);
    PSWIN_CHECK_ARG(valid_dtype(x_dtype) && valid_dtype(out_dtype));
    PSWIN_CHECK_ARG(C % 8 == 0 && aligned16(dout) && aligned16(dx));
    RowGeom g = row_geom(C);
    int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
    long long rows = (long long)B * H * W;
    PSWIN_CHECK_ARG(rows_ok(rows, g.rows_per_block));
    dim3 grid((unsigned)((rows + g.rows_per_block - 1) / g.rows_per_block));
    return dispatch2(out_dtype, x_dtype, [&](auto od, auto xd) {
        hipLaunchKernelGGL((patch_merge_scatter_kernel<decltype(od)::value, decltype(xd)::value>), grid, g.block, 0,
                           (hipStream_t)stream, dout, dx, rows, H, W, H2, W2, g.vpr);
        PSWIN_LAUNCH_RET();
    });
}

extern "C" int pswin_interp_rows(const float* x, const int32_t* idx, const float* wgt, float* out, int B, int S,
                                 int P, int C, void* stream) {
    PSWIN_CHECK_ARG(x && idx && wgt && out && B > 0 && S > 0 && P > 0 && C > 0);
    PSWIN_CHECK_ARG(C % 4 == 0 && aligned16(x) && aligned16(out));
    RowGeom g = row_geom(C);
    long long rows = (long long)B * P;
    PSWIN_CHECK_ARG(rows_ok(rows, g.rows_per_block));
    dim3 grid((unsigned)((rows + g.rows_per_block - 1) / g.rows_per_block));
    hipLaunchKernelGGL(interp_rows_kernel, grid, g.block, 0, (hipStream_t)stream, x, idx, wgt, out, rows, S, P,
                       g.vpr);
    PSWIN_LAUNCH_RET();
}

extern "C" int pswin_interp_rows_adjoint(const float* dout, const int32_t* idx, const float* wgt, float* dx, int B,
                                         int S, int P, int C, void* stream) {
    PSWIN_CHECK_ARG(dout && idx && wgt && dx && B > 0 && S > 0 && P > 0 && C > 0);
    PSWIN_CHECK_ARG(C % 4 == 0 && aligned16(dout) && aligned16(dx));
    RowGeom g = row_geom(C);
    long long rows = (long long)B * P;
    PSWIN_CHECK_ARG(rows_ok(rows, g.rows_per_block));
    dim3 grid((unsigned)((rows + g.rows_per_block - 1) / g.rows_per_block));
    hipLaunchKernelGGL(interp_rows_adjoint_kernel, grid, g.block, 0, (hipStream_t)stream, dout, idx, wgt, dx, rows,
                       S, P, g.vpr);
    PSWIN_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------------------------
// Column sums of a row-major [M, N] matrix (bias gradients: db[n] = sum_m dY[m][n]; split-K partial sums).
// Stage 1: every block streams a strided subset of the rows with 16-byte loads and keeps 8 (bf16) / 4 (f32) fp32
// column sums per thread; the row lanes of a block are combined through LDS.  Stage 2: colsum_kernel over blocks.
// ------------------------------------------------------------------------------------------------------------
namespace {

template <int DT>
__global__ __launch_bounds__(256) void rowsum_partial_kernel(const void* __restrict__ x, long long M, int N,
                                                             int vpr, int vprb, float* __restrict__ part,
                                                             int skip_lo = 0, int skip_hi = 0) {
    constexpr int VE = (DT == PSWIN_BF16) ? 8 : 4;        // elements per 16-byte vector
    __shared__ float red[256 * VE];
    const int rpi = 256 / vprb;                           // rows per iteration of this block
    const int cl = threadIdx.x % vprb, rl = threadIdx.x / vprb;
    const int cg = blockIdx.y * vprb + cl;                // 16-byte column group
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
    // column groups [skip_lo, skip_hi) are known to sum to zero (see pswin_colsum_skip): not read, zeros written
    if (rl < rpi && cg < vpr && !(cg >= skip_lo && cg < skip_hi)) {
        for (long long r = (long long)blockIdx.x * rpi + rl; r < M; r += (long long)gridDim.x * rpi) {
            if constexpr (DT == PSWIN_BF16) {
                const u32x4 raw = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(x) +
                                                                  (size_t)r * N + (size_t)cg * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[2 * e] += __builtin_bit_cast(float, raw[e] << 16);
                    acc[2 * e + 1] += __builtin_bit_cast(float, raw[e] & 0xffff0000u);
                }
            } else {
                const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(x) + (size_t)r * N +
                                                                (size_t)cg * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += v[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) red[threadIdx.x * VE + e] = acc[e];
    __syncthreads();
    if (rl == 0 && cg < vpr) {
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            float s = 0.f;
            for (int q = 0; q < rpi; ++q) s += red[(q * vprb + cl) * VE + e];
            part[(size_t)blockIdx.x * N + (size_t)cg * VE + e] = s;
        }
    }
}

// Short matrices (M <= 1024 rows: split-K partials): one pass.  A block is 32 column groups x 8 row lanes; the row
// lanes are combined through LDS in a fixed order; no second launch.
template <int DT>
__global__ __launch_bounds__(256) void rowsum_direct_kernel(const void* __restrict__ x, int M, int N, int vpr,
                                                            float* __restrict__ out) {
    constexpr int VE = (DT == PSWIN_BF16) ? 8 : 4;
    __shared__ float red[8][32][VE];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int cg = blockIdx.x * 32 + cl;
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
    if (cg < vpr) {
#pragma unroll 4
        for (int r = rl; r < M; r += 8) {
            if constexpr (DT == PSWIN_BF16) {
                const u32x4 raw = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(x) +
                                                                  (size_t)r * N + (size_t)cg * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[2 * e] += __builtin_bit_cast(float, raw[e] << 16);
                    acc[2 * e + 1] += __builtin_bit_cast(float, raw[e] & 0xffff0000u);
                }
            } else {
                const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(x) + (size_t)r * N +
                                                                (size_t)cg * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += v[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) red[rl][cl][e] = acc[e];
    __syncthreads();
    if (rl == 0 && cg < vpr) {
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += red[q][cl][e];
            out[(size_t)cg * VE + e] = s;
        }
    }
}

constexpr int ROWSUM_MAX_BLOCKS = 2048;
constexpr int ROWSUM_DIRECT_MAX_ROWS = 1024;

inline int rowsum_blocks(long long M, int N, int dtype) {
    const int ve = dtype == PSWIN_BF16 ? 8 : 4;
    const int vpr = N / ve;
    const int vprb = vpr < 256 ? vpr : 256;
    const int rpi = 256 / vprb;
    const int ychunks = (vpr + vprb - 1) / vprb;
    long long nb = (M + (long long)rpi * 8 - 1) / ((long long)rpi * 8);        // >= 8 rows per row lane
    const int cap = ROWSUM_MAX_BLOCKS / ychunks > 0 ? ROWSUM_MAX_BLOCKS / ychunks : 1;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return (int)nb;
}

}  // namespace

extern "C" int pswin_colsum_workspace(long long M, int N, int dtype) {
    if (M <= 0 || N <= 0 || !valid_dtype(dtype) || N % 8) return PSWIN_ERR_ARG;
    return rowsum_blocks(M, N, dtype) * N;
}

extern "C" int pswin_colsum(const void* x, int dtype, long long M, int N, float* out, float* workspace, void* stream) {
    PSWIN_CHECK_ARG(x && workspace && M > 0 && N > 0 && valid_dtype(dtype));
    PSWIN_CHECK_ARG(N % 8 == 0 && aligned16(x));
    const int ve = dtype == PSWIN_BF16 ? 8 : 4;
    const int vpr = N / ve;
    if (out && M <= ROWSUM_DIRECT_MAX_ROWS && vpr >= 32 * 64) { // >= 64 blocks of column groups: one pass is enough
        if (dtype == PSWIN_BF16)
            hipLaunchKernelGGL(rowsum_direct_kernel<PSWIN_BF16>, dim3((vpr + 31) / 32), dim3(256), 0,
                               (hipStream_t)stream, x, (int)M, N, vpr, out);
        else
            hipLaunchKernelGGL(rowsum_direct_kernel<PSWIN_F32>, dim3((vpr + 31) / 32), dim3(256), 0,
                               (hipStream_t)stream, x, (int)M, N, vpr, out);
        PSWIN_LAUNCH_RET();
    }
    const int vprb = vpr < 256 ? vpr : 256;
    const int ychunks = (vpr + vprb - 1) / vprb;
    const int blocks = rowsum_blocks(M, N, dtype);
    if (dtype == PSWIN_BF16)
        hipLaunchKernelGGL(rowsum_partial_kernel<PSWIN_BF16>, dim3(blocks, ychunks), dim3(256), 0, (hipStream_t)stream,
                           x, M, N, vpr, vprb, workspace);
    else
        hipLaunchKernelGGL(rowsum_partial_kernel<PSWIN_F32>, dim3(blocks, ychunks), dim3(256), 0, (hipStream_t)stream,
                           x, M, N, vpr, vprb, workspace);
    if (out) launch_colsum(workspace, blocks, N, out, (hipStream_t)stream);     // else: partial rows only
    PSWIN_LAUNCH_RET();
}

extern "C" int pswin_colsum_skip(const void* x, int dtype, long long M, int N, int skip_lo, int skip_hi, float* workspace,
                                 void* stream) {
    PSWIN_CHECK_ARG(x && workspace && M > 0 && N > 0 && valid_dtype(dtype) && N % 8 == 0 && aligned16(x));
    const int ve = dtype == PSWIN_BF16 ? 8 : 4;
    PSWIN_CHECK_ARG(skip_lo >= 0 && skip_lo <= skip_hi && skip_hi <= N && skip_lo % ve == 0 && skip_hi % ve == 0);
    const int vpr = N / ve;
    const int vprb = vpr < 256 ? vpr : 256;
    const int ychunks = (vpr + vprb - 1) / vprb;
    const int blocks = rowsum_blocks(M, N, dtype);
    if (dtype == PSWIN_BF16)
        hipLaunchKernelGGL(rowsum_partial_kernel<PSWIN_BF16>, dim3(blocks, ychunks), dim3(256), 0, (hipStream_t)stream,
                           x, M, N, vpr, vprb, workspace, skip_lo / ve, skip_hi / ve);
    else
        hipLaunchKernelGGL(rowsum_partial_kernel<PSWIN_F32>, dim3(blocks, ychunks), dim3(256), 0, (hipStream_t)stream,
                           x, M, N, vpr, vprb, workspace, skip_lo / ve, skip_hi / ve);
    PSWIN_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------------------------
// Grouped column sums.  A backward pass ends ~120 small "sum these partial rows" reductions (split-K weight-gradient
// partials, bias-gradient partial rows, LayerNorm dgamma/dbeta rows); as separate launches each costs 5-8 us of mostly
// idle GPU.  Their results are only needed once the pass is over, so the host queues them and this kernel runs up to
// 96 of them in ONE launch: the job table travels in the kernel arguments (no device-side table to keep alive, and a
// captured hipGraph holds it by value), every 1024-thread block looks its job up with a binary search over the
// first-block prefix.  Per job the block is shaped G column groups (16 bytes) x RL row lanes, RL in {1, 8, 64} chosen
// from the row count; a row lane adds its rows in ascending order and the lanes are combined in a fixed order.
// ------------------------------------------------------------------------------------------------------------
namespace {

constexpr int RJ_MAX = 96;
struct RJob {
    const void* src;
    float* dst;
    int rows, groups, ld, first_block, lane_shift, dtype;
};
struct RBatch {
    RJob job[RJ_MAX];
    int n;
};
static_assert(sizeof(RBatch) <= 4000, "the job table must fit the 4 KB kernel-argument segment");

__global__ __launch_bounds__(1024) void reduce_jobs_kernel(const RBatch b) {
    __shared__ float red[1024 * 8];
    int lo = 0, hi = b.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (b.job[mid].first_block <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const void* src = b.job[lo].src;
    float* dst = b.job[lo].dst;
    const int rows = b.job[lo].rows, groups = b.job[lo].groups, ld = b.job[lo].ld;
    const int sh = b.job[lo].lane_shift, first = b.job[lo].first_block;
    const bool bf = b.job[lo].dtype == PSWIN_BF16;
    const int RL = 1 << sh, G = 1024 >> sh;
    const int gl = threadIdx.x & (G - 1), rl = threadIdx.x >> (10 - sh);
    const int grp = ((int)blockIdx.x - first) * G + gl;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (grp < groups) {
        const char* p = reinterpret_cast<const char*>(src) + (size_t)grp * 16;
        const size_t stride = (size_t)ld * (bf ? 2 : 4);
        if (bf) {
#pragma unroll 8
            for (int r = rl; r < rows; r += RL) {
                const u32x4 raw = *reinterpret_cast<const u32x4*>(p + (size_t)r * stride);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[2 * e] += __builtin_bit_cast(float, raw[e] << 16);
                    acc[2 * e + 1] += __builtin_bit_cast(float, raw[e] & 0xffff0000u);
                }
            }
        } else {
#pragma unroll 8
            for (int r = rl; r < rows; r += RL) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(p + (size_t)r * stride);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += v[e];
            }
        }
    }
    if (sh != 0) {                                     // uniform per block
#pragma unroll
        for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = acc[e];
        __syncthreads();
        if (sh == 6) {                                 // 64 lanes -> 8: thread (rl < 8, gl) adds lanes rl, rl + 8, ...
            if (rl < 8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float s = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) s += red[((rl + 8 * k) * G + gl) * 8 + e];
                    acc[e] = s;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = acc[e];   // own slot, read by this thread only
            }
            __syncthreads();
        }
        if (rl == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) s += red[(k * G + gl) * 8 + e];
                acc[e] = s;
            }
        }
    }
    if (rl == 0 && grp < groups) {
        if (bf) {
            f32x4 a = {acc[0], acc[1], acc[2], acc[3]}, c = {acc[4], acc[5], acc[6], acc[7]};
            *reinterpret_cast<f32x4*>(dst + (size_t)grp * 8) = a;
            *reinterpret_cast<f32x4*>(dst + (size_t)grp * 8 + 4) = c;
        } else {
            f32x4 a = {acc[0], acc[1], acc[2], acc[3]};
            *reinterpret_cast<f32x4*>(dst + (size_t)grp * 4) = a;
        }
    }
}

}  // namespace

extern "C" int pswin_reduce_jobs(const pswin_reduce_job* jobs, int n_jobs, void* stream) {
    PSWIN_CHECK_ARG(jobs && n_jobs > 0);
    for (int j = 0; j < n_jobs; ++j) {
        const pswin_reduce_job& q = jobs[j];
        const int ve = q.dtype == PSWIN_BF16 ? 8 : 4;
        PSWIN_CHECK_ARG(q.src && q.dst && valid_dtype(q.dtype) && q.rows > 0 && q.cols > 0 && q.cols % ve == 0);
        PSWIN_CHECK_ARG(q.ld >= q.cols && q.ld % ve == 0 && aligned16(q.src) && aligned16(q.dst));
    }
    for (int at = 0; at < n_jobs; at += RJ_MAX) {
        RBatch b;
        b.n = n_jobs - at < RJ_MAX ? n_jobs - at : RJ_MAX;
        long long blocks = 0;
        for (int j = 0; j < b.n; ++j) {
            const pswin_reduce_job& q = jobs[at + j];
            RJob& r = b.job[j];
            r.src = q.src;
            r.dst = q.dst;
            r.rows = q.rows;
            r.groups = q.cols / (q.dtype == PSWIN_BF16 ? 8 : 4);
            r.ld = q.ld;
            r.dtype = q.dtype;
            r.lane_shift = q.rows <= 16 ? 0 : (q.rows <= 128 ? 3 : 6);
            r.first_block = (int)blocks;
            const int G = 1024 >> r.lane_shift;
            blocks += (r.groups + G - 1) / G;
        }
        PSWIN_CHECK_ARG(blocks < 0x7fffffffll);
        hipLaunchKernelGGL(reduce_jobs_kernel, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, b);
    }
    PSWIN_LAUNCH_RET();
}
