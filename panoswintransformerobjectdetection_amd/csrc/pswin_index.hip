// Index maps of the PanoSwin window layout (integer work, bit-exact against the reference).
//
// Replaces, for the gather/scatter kernels, the chain of full-tensor copies the reference makes per block:
// torch.roll / flip / cat (WindowTransition, HOT:326-409), F.pad (HOT:486-491), window_partition /
// window_reverse (HOT:64-92) and the mask construction of BasicLayer._get_attention_mask (HOT:664-688).
// HOT = mmdet/models/backbones/simple_panoswin_transformer.py of the reference.
#include "pswin_common.hpp"

using namespace pswin;

namespace {

struct Grid {
    int Hp, Wp, SH, SW, Wq, nWx, nW;
};

__host__ __device__ inline Grid make_grid(int mode, int H, int W) {
    Grid g;
    if (mode == PSWIN_MODE_PANO) {
        g.Wq = W + (W & 1);  // ew2ns right-pads one zero column when W is odd (HOT:344-347)
        g.SH = 2 * H;
        g.SW = g.Wq / 2;
    } else {
        g.Wq = W;
        g.SH = H;
        g.SW = W;
    }
    g.Hp = ceil_to(g.SH, PSWIN_WS);
    g.Wp = ceil_to(g.SW, PSWIN_WS);
    g.nWx = g.Wp / PSWIN_WS;
    g.nW = (g.Hp / PSWIN_WS) * g.nWx;
    return g;
}

// source token of padded-grid position (Y, X), or -1
__device__ inline int source_token(int mode, int H, int W, int shift, const Grid& g, int Y, int X) {
    if (mode == PSWIN_MODE_PANO) {
        if (Y >= g.SH || X >= g.SW) return -1;
        int y = ((Y - shift) % g.SH + g.SH) % g.SH;  // roll H by +shift on the north-south layout (HOT:406)
        int h, w1;
        if (y < H) {                      // top half = right half of the east-west map, flipped in H and W
            h = H - 1 - y;
            w1 = g.Wq - 1 - X;
        } else {                          // bottom half = left half, as is
            h = y - H;
            w1 = X;
        }
        if (w1 >= W) return -1;           // the zero column of an odd-width map
        int w = ((w1 - shift) % W + W) % W;  // roll W by +shift before the fold (HOT:400)
        return h * W + w;
    }
    // planar: zero-pad to (Hp, Wp) first, then roll by (-shift, -shift) (HOT:522-523)
    int sy = Y + shift;
    if (sy >= g.Hp) sy -= g.Hp;
    int sx = X + shift;
    if (sx >= g.Wp) sx -= g.Wp;
    if (sy >= H || sx >= W) return -1;
    return sy * W + sx;
}

__global__ void window_map_kernel(int mode, int H, int W, int shift, Grid g, int32_t* __restrict__ map,
                                  int32_t* __restrict__ inv) {
    int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= g.nW * PSWIN_WTOK) return;
    int win = slot / PSWIN_WTOK, tok = slot - win * PSWIN_WTOK;
    int Y = (win / g.nWx) * PSWIN_WS + tok / PSWIN_WS;
    int X = (win % g.nWx) * PSWIN_WS + tok % PSWIN_WS;
    int src = source_token(mode, H, W, shift, g, Y, X);
    map[slot] = src;
    if (src >= 0 && inv) inv[src] = slot;
}

__device__ inline int region(int v, int L, int shift) { return (v >= L - PSWIN_WS) + (v >= L - shift); }

__global__ void planar_mask_kernel(int shift, Grid g, float* __restrict__ mask) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    int total = g.nW * PSWIN_WTOK * PSWIN_WTOK;
    if (e >= total) return;
    int win = e / (PSWIN_WTOK * PSWIN_WTOK);
    int r = e - win * PSWIN_WTOK * PSWIN_WTOK;
    int i = r / PSWIN_WTOK, j = r - i * PSWIN_WTOK;
    int y0 = (win / g.nWx) * PSWIN_WS, x0 = (win % g.nWx) * PSWIN_WS;
    int ri = 3 * region(y0 + i / PSWIN_WS, g.Hp, shift) + region(x0 + i % PSWIN_WS, g.Wp, shift);
    int rj = 3 * region(y0 + j / PSWIN_WS, g.Hp, shift) + region(x0 + j % PSWIN_WS, g.Wp, shift);
    mask[e] = (ri != rj) ? -100.0f : 0.0f;
}

}  // namespace

extern "C" int pswin_version(void) { return PSWIN_ABI_VERSION; }

extern "C" int pswin_window_grid(int mode, int H, int W, int* Hp, int* Wp, int* n_windows) {
    PSWIN_CHECK_ARG(mode == PSWIN_MODE_PANO || mode == PSWIN_MODE_PLANAR);
    PSWIN_CHECK_ARG(H > 0 && W > 0);
    Grid g = make_grid(mode, H, W);
    if (Hp) *Hp = g.Hp;
    if (Wp) *Wp = g.Wp;
    if (n_windows) *n_windows = g.nW;
    return PSWIN_OK;
}

extern "C" int pswin_window_map(int mode, int H, int W, int shift, int32_t* map, int32_t* inv, void* stream) {
    PSWIN_CHECK_ARG(mode == PSWIN_MODE_PANO || mode == PSWIN_MODE_PLANAR);
    PSWIN_CHECK_ARG(H > 0 && W > 0 && map != nullptr);
    PSWIN_CHECK_ARG(shift >= 0 && shift < PSWIN_WS);
    PSWIN_CHECK_ARG((long long)H * W < (1ll << 30));
    Grid g = make_grid(mode, H, W);
    int n = g.nW * PSWIN_WTOK;
    hipLaunchKernelGGL(window_map_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, mode, H, W,
                       shift, g, map, inv);
    PSWIN_LAUNCH_RET();
}

extern "C" int pswin_planar_mask(int H, int W, int shift, float* mask, void* stream) {
    PSWIN_CHECK_ARG(H > 0 && W > 0 && mask != nullptr);
    PSWIN_CHECK_ARG(shift > 0 && shift < PSWIN_WS);
    Grid g = make_grid(PSWIN_MODE_PLANAR, H, W);
    long long n = (long long)g.nW * PSWIN_WTOK * PSWIN_WTOK;
    PSWIN_CHECK_ARG(n < (1ll << 31));
    hipLaunchKernelGGL(planar_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       shift, g, mask);
    PSWIN_LAUNCH_RET();
}
