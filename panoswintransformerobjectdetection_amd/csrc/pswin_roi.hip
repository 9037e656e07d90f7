// RoIAlign over an FPN pyramid (gfx950): the RoI extractors of the Mask R-CNN step that calls the backbone.
//
// configs/_base_/models/mask_rcnn_swin_fpn.py:44-48, 63-67: SingleRoIExtractor(RoIAlign(output_size 7 / 14, sampling_ratio = 0),
// featmap_strides 4..32); mmdet/models/roi_heads/roi_extractors/single_level_roi_extractor.py:78-108: every RoI is pooled from ONE
// level (the caller maps it, :55-60).  RoIAlign itself lives in the un-vendored mmcv.ops, so what is implemented is the published
// algorithm of that operator (Mask R-CNN / detectron / torchvision / mmcv, `aligned` = mmcv's default True):
//   roi * spatial_scale - 0.5; bins of (roi_h / P) x (roi_w / P); per bin an ADAPTIVE grid of ceil(roi_h / P) x ceil(roi_w / P)
//   samples (sampling_ratio = 0) at the sub-bin centres; a sample outside [-1, H] x [-1, W] contributes 0, otherwise the
//   coordinate is clamped to [0, H-1] and read bilinearly; the bin is the mean over the grid.
// PARITY: unpinned against the reference (no mmcv.ops source or fixture in the tree); tests check it against a plain PyTorch
// statement of the same definition.
//
// Layout: feature maps NHWC (a pixel's C channels are one contiguous 512-byte row for C = 256 bf16), output [R, P, P, C].
// Forward: one wave per (RoI, bin); a pixel row is read by C / VEC lanes with 16-byte loads, the 64 / (C / VEC) lane groups of the
// wave take different samples of the bin and are summed at the end.  A launch is R * P * P waves' worth of independent bins:
// 50,176 for either head of the step (1024 RoIs x 49, 256 x 196).
// Backward: the adjoint scatter with f32 atomics, one per touched cell of a bin (separable weights), (one dword per lane, 64 consecutive channels per wave instruction = the 256
// contiguous bytes the memory-side atomic units take at full rate; MI355X_MICROARCH.md, Global float atomics) into f32 NHWC
// gradient maps that the caller zeroes.  Sums over RoIs therefore depend on arrival order (as torch's grid_sample backward,
// which this replaces): not bitwise reproducible from run to run.
#include "pswin_common.hpp"

using namespace pswin;

namespace {

constexpr int MAXL = PSWIN_ROI_MAX_LEVELS;
struct Levels {
    const void* feat[MAXL];     // forward: NHWC feature maps
    float* dfeat[MAXL];         // backward: NHWC f32 gradient maps (accumulated into)
    int H[MAXL], W[MAXL];
    float scale[MAXL];          // 1 / stride
    int n;
};

struct BinGeom {
    int b, lvl, H, W, gh, gw;
    float y0, x0, sy, sx, inv_count;    // sample (iy, ix) sits at (y0 + (iy + .5) sy, x0 + (ix + .5) sx)
};

__device__ inline BinGeom bin_geom(const Levels& lv, const float* __restrict__ rois, const int* __restrict__ roi_level, int r, int ph, int pw,
                                   int P, int sampling_ratio, int aligned) {
    BinGeom g;
    const float* q = rois + 5 * (size_t)r;
    g.b = (int)q[0];
    int l = roi_level[r];
    l = l < 0 ? 0 : (l >= lv.n ? lv.n - 1 : l);
    g.lvl = l;
    g.H = lv.H[l];
    g.W = lv.W[l];
    const float s = lv.scale[l], off = aligned ? 0.5f : 0.f;
    const float rsw = q[1] * s - off, rsh = q[2] * s - off, rew = q[3] * s - off, reh = q[4] * s - off;
    float rw = rew - rsw, rh = reh - rsh;
    if (!aligned) {
        rw = fmaxf(rw, 1.f);
        rh = fmaxf(rh, 1.f);
    }
    const float bh = rh / (float)P, bw = rw / (float)P;
    g.gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)P);
    g.gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)P);
    g.gh = g.gh < 0 ? 0 : g.gh;
    g.gw = g.gw < 0 ? 0 : g.gw;
    const int cnt = g.gh * g.gw;
    g.inv_count = 1.f / (float)(cnt > 0 ? cnt : 1);
    g.sy = g.gh > 0 ? bh / (float)g.gh : 0.f;
    g.sx = g.gw > 0 ? bw / (float)g.gw : 0.f;
    g.y0 = rsh + (float)ph * bh;
    g.x0 = rsw + (float)pw * bw;
    return g;
}

struct Taps {
    int yl, yh, xl, xh;
    float w1, w2, w3, w4;
    bool in;
};
__device__ inline Taps taps_of(float y, float x, int H, int W) {
    Taps t;
    t.in = !(y < -1.f || y > (float)H || x < -1.f || x > (float)W);
    y = fmaxf(y, 0.f);
    x = fmaxf(x, 0.f);
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= H - 1) {
        yh = yl = H - 1;
        y = (float)yl;
    } else {
        yh = yl + 1;
    }
    if (xl >= W - 1) {
        xh = xl = W - 1;
        x = (float)xl;
    } else {
        xh = xl + 1;
    }
    const float ly = y - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
    t.yl = yl; t.yh = yh; t.xl = xl; t.xh = xh;
    t.w1 = hy * hx; t.w2 = hy * lx; t.w3 = ly * hx; t.w4 = ly * lx;
    return t;
}

constexpr int BINS_PER_WAVE = 4, RTHREADS = 256;

// VEC channels per lane (16 bytes): 8 bf16 / 4 f32; LPP = C / VEC lanes per pixel (a power of two <= 64)
template <int DT>
__global__ __launch_bounds__(RTHREADS) void roi_align_fwd_kernel(const Levels lv, const float* __restrict__ rois, const int* __restrict__ roi_level,
                                                                 long long nbins, int C, int P, int sampling_ratio, int aligned, void* __restrict__ out) {
    constexpr int VEC = DT == PSWIN_BF16 ? 8 : 4;
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (RTHREADS / 64) + (threadIdx.x >> 6);
    const int lpp = C / VEC, groups = 64 / lpp;
    const int grp = lane / lpp, ch = (lane - grp * lpp) * VEC;
    const int PP = P * P;
    for (int k = 0; k < BINS_PER_WAVE; ++k) {
        const long long idx = wave * BINS_PER_WAVE + k;
        if (idx >= nbins) return;                       // wave-uniform
        const int r = (int)(idx / PP), bin = (int)(idx - (long long)r * PP);
        const int ph = bin / P, pw = bin - ph * P;
        const BinGeom g = bin_geom(lv, rois, roi_level, r, ph, pw, P, sampling_ratio, aligned);
        const char* base = reinterpret_cast<const char*>(lv.feat[g.lvl]);
        const size_t esz = DT == PSWIN_BF16 ? 2 : 4;
        const size_t img = (size_t)g.b * g.H * g.W;
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        for (int iy = 0; iy < g.gh; ++iy) {
            const float y = g.y0 + ((float)iy + 0.5f) * g.sy;
            for (int ix = grp; ix < g.gw; ix += groups) {
                const float x = g.x0 + ((float)ix + 0.5f) * g.sx;
                const Taps t = taps_of(y, x, g.H, g.W);
                if (!t.in) continue;
                const size_t o1 = ((img + (size_t)t.yl * g.W + t.xl) * C + ch) * esz, o2 = ((img + (size_t)t.yl * g.W + t.xh) * C + ch) * esz;
                const size_t o3 = ((img + (size_t)t.yh * g.W + t.xl) * C + ch) * esz, o4 = ((img + (size_t)t.yh * g.W + t.xh) * C + ch) * esz;
                const u32x4 a = *reinterpret_cast<const u32x4*>(base + o1), b = *reinterpret_cast<const u32x4*>(base + o2);
                const u32x4 c = *reinterpret_cast<const u32x4*>(base + o3), d = *reinterpret_cast<const u32x4*>(base + o4);
                if constexpr (DT == PSWIN_BF16) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[2 * e] += t.w1 * __builtin_bit_cast(float, a[e] << 16) + t.w2 * __builtin_bit_cast(float, b[e] << 16) +
                                      t.w3 * __builtin_bit_cast(float, c[e] << 16) + t.w4 * __builtin_bit_cast(float, d[e] << 16);
                        acc[2 * e + 1] += t.w1 * __builtin_bit_cast(float, a[e] & 0xffff0000u) + t.w2 * __builtin_bit_cast(float, b[e] & 0xffff0000u) +
                                          t.w3 * __builtin_bit_cast(float, c[e] & 0xffff0000u) + t.w4 * __builtin_bit_cast(float, d[e] & 0xffff0000u);
                    }
                } else {
                    const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b), cf = __builtin_bit_cast(f32x4, c),
                                df = __builtin_bit_cast(f32x4, d);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += t.w1 * af[e] + t.w2 * bf[e] + t.w3 * cf[e] + t.w4 * df[e];
                }
            }
        }
        // sum over the lane groups (groups is a power of two; every lane takes part: the loop bounds above are group-uniform or
        // end in a full-wave reconvergence before this point)
        for (int m = lpp; m < 64; m <<= 1)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] += __shfl_xor(acc[e], m, 64);
        if (grp == 0) {
            char* o = reinterpret_cast<char*>(out) + ((size_t)idx * C + ch) * esz;
            if constexpr (DT == PSWIN_BF16) {
                u32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = pack2_bf16(acc[2 * e] * g.inv_count, acc[2 * e + 1] * g.inv_count);
                *reinterpret_cast<u32x4*>(o) = v;
            } else {
                *reinterpret_cast<f32x4*>(o) = f32x4{acc[0] * g.inv_count, acc[1] * g.inv_count, acc[2] * g.inv_count, acc[3] * g.inv_count};
            }
        }
    }
}

// lane l owns channels l, l + 64, ... (C / 64 of them): every atomic wave instruction adds 256 contiguous bytes
template <int DT, int CK>
__global__ __launch_bounds__(RTHREADS) void roi_align_bwd_kernel(const Levels lv, const float* __restrict__ rois, const int* __restrict__ roi_level,
                                                                 long long nbins, int P, int sampling_ratio, int aligned, const void* __restrict__ dout) {
    constexpr int C = 64 * CK;
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (RTHREADS / 64) + (threadIdx.x >> 6);
    const int PP = P * P;
    for (int k = 0; k < BINS_PER_WAVE; ++k) {
        const long long idx = wave * BINS_PER_WAVE + k;
        if (idx >= nbins) return;
        const int r = (int)(idx / PP), bin = (int)(idx - (long long)r * PP);
        const int ph = bin / P, pw = bin - ph * P;
        const BinGeom g = bin_geom(lv, rois, roi_level, r, ph, pw, P, sampling_ratio, aligned);
        float gv[CK];
#pragma unroll
        for (int j = 0; j < CK; ++j) {
            const size_t o = (size_t)idx * C + lane + 64 * j;
            if constexpr (DT == PSWIN_BF16) gv[j] = bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(dout)[o]) * g.inv_count;
            else gv[j] = reinterpret_cast<const float*>(dout)[o] * g.inv_count;
        }
        float* base = lv.dfeat[g.lvl] + (size_t)g.b * g.H * g.W * C + lane;
        // The bilinear weight of a sample on a cell is (row weight) x (column weight) and "in range" is (row in range) and (column in
        // range), so the sum over the bin's gh x gw samples factors:  w(cell r, c) = (sum_i wy_i(r)) (sum_j wx_j(c)).  One atomic per
        // touched CELL -- at most (gh + 1)(gw + 1) -- instead of four per SAMPLE: 9 instead of 16 for a 2 x 2 grid, 25 instead of 64 for
        // 4 x 4; the kernel is bound by the memory-side atomic rate (0.93 ms per launch in the Mask R-CNN step before this).  Lane k <
        // 16 accumulates the weight of footprint row / column k; footprints wider than 16 cells take the per-sample path below.
        const int kcell = lane & 15;
        int fy = 0, ly_ = -1, fx = 0, lx_ = -1;
        float wy = 0.f, wx = 0.f;
        bool anyy = false, anyx = false;
        // (samples are visited in the order of increasing coordinate -- an inverted box with a fixed grid has a negative step -- so that
        // the first one in range is the footprint's origin)
        for (int i0 = 0; i0 < g.gh; ++i0) {
            const int i = g.sy >= 0.f ? i0 : g.gh - 1 - i0;
            float p = g.y0 + ((float)i + 0.5f) * g.sy;
            if (p < -1.f || p > (float)g.H) continue;
            p = fmaxf(p, 0.f);
            int lo = (int)p, hi;
            if (lo >= g.H - 1) {
                hi = lo = g.H - 1;
                p = (float)lo;
            } else {
                hi = lo + 1;
            }
            const float l = p - (float)lo, h = 1.f - l;
            if (!anyy) {
                fy = lo;
                anyy = true;
            }
            ly_ = hi;
            wy += (lo - fy == kcell ? h : 0.f) + (hi - fy == kcell ? l : 0.f);
        }
        for (int i0 = 0; i0 < g.gw; ++i0) {
            const int i = g.sx >= 0.f ? i0 : g.gw - 1 - i0;
            float p = g.x0 + ((float)i + 0.5f) * g.sx;
            if (p < -1.f || p > (float)g.W) continue;
            p = fmaxf(p, 0.f);
            int lo = (int)p, hi;
            if (lo >= g.W - 1) {
                hi = lo = g.W - 1;
                p = (float)lo;
            } else {
                hi = lo + 1;
            }
            const float l = p - (float)lo, h = 1.f - l;
            if (!anyx) {
                fx = lo;
                anyx = true;
            }
            lx_ = hi;
            wx += (lo - fx == kcell ? h : 0.f) + (hi - fx == kcell ? l : 0.f);
        }
        // the geometry is the same in every lane: counts and origins as scalars
        const int ny = __builtin_amdgcn_readfirstlane(anyy ? ly_ - fy + 1 : 0), nx = __builtin_amdgcn_readfirstlane(anyx ? lx_ - fx + 1 : 0);
        const int r0 = __builtin_amdgcn_readfirstlane(fy), c0 = __builtin_amdgcn_readfirstlane(fx);
        if (ny <= 16 && nx <= 16) {
            for (int cy = 0; cy < ny; ++cy) {
                const float wr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wy), cy));
                if (wr == 0.f) continue;
                float* rowp = base + ((size_t)(r0 + cy) * g.W + c0) * C;
                for (int cx = 0; cx < nx; ++cx) {
                    const float w = wr * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wx), cx));
                    if (w == 0.f) continue;
                    float* pc = rowp + (size_t)cx * C;
#pragma unroll
                    for (int j = 0; j < CK; ++j) atomicAdd(pc + 64 * j, w * gv[j]);
                }
            }
            continue;
        }
        for (int iy = 0; iy < g.gh; ++iy) {
            const float y = g.y0 + ((float)iy + 0.5f) * g.sy;
            for (int ix = 0; ix < g.gw; ++ix) {
                const float x = g.x0 + ((float)ix + 0.5f) * g.sx;
                const Taps t = taps_of(y, x, g.H, g.W);
                if (!t.in) continue;                    // wave-uniform (the geometry is the same in every lane)
                float* p1 = base + ((size_t)t.yl * g.W + t.xl) * C;
                float* p2 = base + ((size_t)t.yl * g.W + t.xh) * C;
                float* p3 = base + ((size_t)t.yh * g.W + t.xl) * C;
                float* p4 = base + ((size_t)t.yh * g.W + t.xh) * C;
#pragma unroll
                for (int j = 0; j < CK; ++j) {
                    atomicAdd(p1 + 64 * j, t.w1 * gv[j]);
                    atomicAdd(p2 + 64 * j, t.w2 * gv[j]);
                    atomicAdd(p3 + 64 * j, t.w3 * gv[j]);
                    atomicAdd(p4 + 64 * j, t.w4 * gv[j]);
                }
            }
        }
    }
}

// ---- greedy NMS over groups of score-sorted boxes (the proposal stage of the RPN: one group per image and pyramid level) ----------
// keep[i] = no kept j < i with IoU(i, j) > thr -- the sequential rule (mmcv.ops.nms semantics; IoU as detector.box_iou: clamped
// widths, union floored at 1e-6).  Two launches: (1) the suppression bit masks mask[g][i][w] (bit j of word w: IoU(i, 64 w + j) > thr,
// columns right of the row only), one 64-thread workgroup per (64 rows, one word, group) -- n^2 / 2 IoUs spread over the whole chip;
// (2) one wave per group walks the rows in order: lane w owns word w of the "removed" set, a row's verdict is one v_readlane, its
// mask row one OR; the rows' words are fetched 16 rows ahead.  (A single-workgroup-per-group version that also built the masks took
// 639 us per launch: 2 M IoUs on one CU.)  Replaces a fixed-point iteration of twelve [n] x [n, n] matrix-vector sweeps per group
// (120 GEMV + 240 element-wise launches per Mask R-CNN step).
constexpr int NMS_MAX = 2048, NMS_WORDS = NMS_MAX / 64;

__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, const int* __restrict__ counts, int nmax, float thr,
                                                      unsigned long long* __restrict__ mask) {
    __shared__ f32x4 cb[64];
    __shared__ float ca[64];
    const int w = blockIdx.x, rc = blockIdx.y, gidx = blockIdx.z, r = threadIdx.x;
    if (w < rc) return;                                // columns left of the rows: never read
    int n = counts ? counts[gidx] : nmax;
    n = n < 0 ? 0 : (n > nmax ? nmax : n);
    if (64 * rc >= n) return;
    const f32x4* src = reinterpret_cast<const f32x4*>(boxes) + (size_t)gidx * nmax;
    const int j = 64 * w + r, i = 64 * rc + r;
    if (j < n) {
        const f32x4 b = src[j];
        cb[r] = b;
        ca[r] = fmaxf(b[2] - b[0], 0.f) * fmaxf(b[3] - b[1], 0.f);
    }
    __syncthreads();
    unsigned long long bits = 0ull;
    if (i < n) {
        const f32x4 a = src[i];
        const float aa = fmaxf(a[2] - a[0], 0.f) * fmaxf(a[3] - a[1], 0.f);
        const int jn = n - 64 * w < 64 ? n - 64 * w : 64;
        for (int jj = 0; jj < jn; ++jj) {
            const f32x4 b = cb[jj];                     // the same address in every lane: an LDS broadcast
            const float iw = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]), 0.f), ih = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]), 0.f);
            const float inter = iw * ih;
            const float iou = inter / fmaxf(aa + ca[jj] - inter, 1e-6f);
            if (64 * w + jj > i && iou > thr) bits |= 1ull << jj;
        }
    }
    mask[((size_t)gidx * nmax + i) * NMS_WORDS + w] = bits;      // rows >= n of a started chunk: zeros (i < nmax: chunks are whole)
}

__global__ __launch_bounds__(64) void nms_scan_kernel(const int* __restrict__ counts, int nmax, const unsigned long long* __restrict__ mask,
                                                      unsigned char* __restrict__ keep) {
    const int gidx = blockIdx.x, lane = threadIdx.x;
    int n = counts ? counts[gidx] : nmax;
    n = n < 0 ? 0 : (n > nmax ? nmax : n);
    unsigned char* kp = keep + (size_t)gidx * nmax;
    for (int i = lane; i < nmax; i += 64) kp[i] = 0;
    const int words = (n + 63) >> 6;
    const unsigned long long* mg = mask + (size_t)gidx * nmax * NMS_WORDS;
    unsigned long long remv = 0ull;                    // lane w holds word w of the removed set
    // rows are fetched 16 at a time, one batch ahead of the walk (a batch is 16 x 256 B of mask words: without the prefetch every batch
    // cost a memory round trip, 190 of the kernel's 245 us); batch t = rows 16 t .. 16 t + 15, chunk c = t >> 2
    const int batches = 4 * words;
    auto fetch = [&](int t, unsigned long long (&m)[16]) {
        const int c = t >> 2;
#pragma unroll
        for (int r = 0; r < 16; ++r) m[r] = (t < batches && lane >= c && lane < words) ? mg[(size_t)(16 * t + r) * NMS_WORDS + lane] : 0ull;
    };
    auto lane64 = [&](unsigned long long v, int l) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffull), l);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
        return ((unsigned long long)hi << 32) | lo;
    };
    unsigned long long cur = 0ull;                     // word c of the removed set as a scalar: the only word the chunk's verdicts read
    unsigned long long kw = 0ull;                      // the chunk's verdicts, written once per chunk by all lanes: a store inside the
                                                       // (data-dependent) branch makes the wait for the prefetched rows a vmcnt(0)
    auto walk = [&](int t, const unsigned long long (&m)[16]) {
        const int c = t >> 2;
        if ((t & 3) == 0) {
            cur = lane64(remv, c);
            kw = 0ull;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = 16 * (t & 3) + r, i = 16 * t + r;
            if (i < n && !((cur >> rr) & 1ull)) {       // wave-uniform
                remv |= m[r];
                cur |= lane64(m[r], c);
                kw |= 1ull << rr;
            }
        }
        if ((t & 3) == 3) kp[64 * c + lane] = (unsigned char)((kw >> lane) & 1ull);       // 64 c + lane < nmax: whole chunks
    };
    unsigned long long ma[16], mb[16];
    fetch(0, ma);
    for (int t = 0; t < batches; t += 2) {
        fetch(t + 1, mb);
        walk(t, ma);
        fetch(t + 2, ma);
        walk(t + 1, mb);
    }
}

int fill_levels(Levels& lv, const pswin_roi_levels* in, bool bwd, int C, int dtype) {
    PSWIN_CHECK_ARG(in && in->n_levels >= 1 && in->n_levels <= MAXL);
    lv.n = in->n_levels;
    for (int l = 0; l < MAXL; ++l) {
        const bool on = l < lv.n;
        lv.feat[l] = on ? in->feat[l] : nullptr;
        lv.dfeat[l] = on ? in->dfeat[l] : nullptr;
        lv.H[l] = on ? in->H[l] : 1;
        lv.W[l] = on ? in->W[l] : 1;
        lv.scale[l] = on ? in->spatial_scale[l] : 1.f;
        if (on) {
            PSWIN_CHECK_ARG(lv.H[l] > 0 && lv.W[l] > 0 && lv.scale[l] > 0.f);
            PSWIN_CHECK_ARG(bwd ? (lv.dfeat[l] && aligned16(lv.dfeat[l])) : (lv.feat[l] && aligned16(lv.feat[l])));
        }
    }
    (void)C;
    (void)dtype;
    return PSWIN_OK;
}

}  // namespace

extern "C" {

int pswin_roi_align_supported(int C, int dtype) {
    if (!valid_dtype(dtype) || C < 64 || C % 64 || C > 512) return 0;
    const int lpp = C / (dtype == PSWIN_BF16 ? 8 : 4);
    return lpp <= 64 && (lpp & (lpp - 1)) == 0;
}

int pswin_roi_align_fwd(const pswin_roi_levels* levels, const float* rois, const int32_t* roi_level, int R, int C, int P, int sampling_ratio,
                        int aligned, int dtype, void* out, void* stream) {
    PSWIN_CHECK_ARG(rois && roi_level && out && R > 0 && P > 0 && P <= 64 && sampling_ratio >= 0 && aligned16(out));
    if (!pswin_roi_align_supported(C, dtype)) return PSWIN_ERR_UNSUPPORTED;
    Levels lv;
    if (const int rc = fill_levels(lv, levels, false, C, dtype)) return rc;
    const long long nbins = (long long)R * P * P;
    const long long per_wg = (RTHREADS / 64) * BINS_PER_WAVE;
    const unsigned grid = (unsigned)((nbins + per_wg - 1) / per_wg);
    if (dtype == PSWIN_BF16)
        hipLaunchKernelGGL(roi_align_fwd_kernel<PSWIN_BF16>, dim3(grid), dim3(RTHREADS), 0, (hipStream_t)stream, lv, rois, roi_level, nbins, C, P,
                           sampling_ratio, aligned, out);
    else
        hipLaunchKernelGGL(roi_align_fwd_kernel<PSWIN_F32>, dim3(grid), dim3(RTHREADS), 0, (hipStream_t)stream, lv, rois, roi_level, nbins, C, P,
                           sampling_ratio, aligned, out);
    PSWIN_LAUNCH_RET();
}

int pswin_roi_align_bwd(const pswin_roi_levels* levels, const float* rois, const int32_t* roi_level, int R, int C, int P, int sampling_ratio,
                        int aligned, int dtype, const void* dout, void* stream) {
    PSWIN_CHECK_ARG(rois && roi_level && dout && R > 0 && P > 0 && P <= 64 && sampling_ratio >= 0);
    if (!pswin_roi_align_supported(C, dtype)) return PSWIN_ERR_UNSUPPORTED;
    Levels lv;
    if (const int rc = fill_levels(lv, levels, true, C, dtype)) return rc;
    const long long nbins = (long long)R * P * P;
    const long long per_wg = (RTHREADS / 64) * BINS_PER_WAVE;
    const unsigned grid = (unsigned)((nbins + per_wg - 1) / per_wg);
#define PSWIN_ROI_BWD(DT, CK)                                                                                                              \
    hipLaunchKernelGGL((roi_align_bwd_kernel<DT, CK>), dim3(grid), dim3(RTHREADS), 0, (hipStream_t)stream, lv, rois, roi_level, nbins, P, \
                       sampling_ratio, aligned, dout)
    const int ck = C / 64;
    if (dtype == PSWIN_BF16) {
        switch (ck) {
            case 1: PSWIN_ROI_BWD(PSWIN_BF16, 1); break;
            case 2: PSWIN_ROI_BWD(PSWIN_BF16, 2); break;
            case 4: PSWIN_ROI_BWD(PSWIN_BF16, 4); break;
            case 8: PSWIN_ROI_BWD(PSWIN_BF16, 8); break;
            default: return PSWIN_ERR_UNSUPPORTED;
        }
    } else {
        switch (ck) {
            case 1: PSWIN_ROI_BWD(PSWIN_F32, 1); break;
            case 2: PSWIN_ROI_BWD(PSWIN_F32, 2); break;
            case 4: PSWIN_ROI_BWD(PSWIN_F32, 4); break;
            default: return PSWIN_ERR_UNSUPPORTED;
        }
    }
#undef PSWIN_ROI_BWD
    PSWIN_LAUNCH_RET();
}

/* Greedy NMS over `groups` independent lists of boxes sorted by descending score (detector stand-in of mmcv.ops.nms inside
 * RPNHead.get_bboxes): boxes f32 [groups][nmax][4] (x1, y1, x2, y2), counts int32 [groups] (valid boxes per group, or NULL = nmax),
 * keep uint8 [groups][nmax] (1 = kept; entries past a group's count are 0), workspace: pswin_nms_workspace(groups, nmax) bytes.
 * nmax <= 2048 and a multiple of 64. */
int pswin_nms_workspace(int groups, int nmax) {
    if (groups <= 0 || groups > 2048 || nmax <= 0 || nmax > NMS_MAX || nmax % 64) return PSWIN_ERR_ARG;
    return groups * nmax * NMS_WORDS * 8;             // <= 1 GiB
}

int pswin_nms_groups(const float* boxes, const int32_t* counts, int groups, int nmax, float iou_threshold, unsigned char* keep, void* workspace,
                     void* stream) {
    PSWIN_CHECK_ARG(boxes && keep && workspace && groups > 0 && groups <= 2048 && nmax > 0 && nmax <= NMS_MAX && nmax % 64 == 0 && iou_threshold >= 0.f);
    PSWIN_CHECK_ARG(aligned16(boxes) && aligned16(workspace));
    const int chunks = nmax / 64;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(chunks, chunks, groups), dim3(64), 0, (hipStream_t)stream, boxes, counts, nmax, iou_threshold,
                       reinterpret_cast<unsigned long long*>(workspace));
    hipLaunchKernelGGL(nms_scan_kernel, dim3(groups), dim3(64), 0, (hipStream_t)stream, counts, nmax,
                       reinterpret_cast<const unsigned long long*>(workspace), keep);
    PSWIN_LAUNCH_RET();
}

}  // extern "C"
