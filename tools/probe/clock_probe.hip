// Clock held by the chip under the fused window-attention kernel (not part of the library): pswin_fused.hip built with
// -DPSWIN_FUSED_CLOCK_PROBE stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around each workgroup's window loop.
// The kernel is launched back to back on random data for ~2 s first (the DVFS state of a burst is not the state of a run), then the
// median over the workgroups of  d(s_memtime) / d(s_memrealtime) x 100 MHz  is printed next to the launch time and the algorithmic
// TFLOP/s:   algorithmic FLOP per window = 2 * 49 * 96 * 288 + 3 * 4 * 49 * 49 * 32 + 2 * 49 * 96 * 96  (SURVEY 8d, C = 96).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -DPSWIN_FUSED_CLOCK_PROBE -I include -I panoswintransformerobjectdetection_amd/csrc \
//         tools/probe/clock_probe.hip -o tools/probe/bin/clock_probe
#include "../../panoswintransformerobjectdetection_amd/csrc/pswin_fused.hip"
#include <algorithm>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>

static void fill_bf16(unsigned short* d, size_t n, float scale) {
    std::vector<unsigned short> h(n);
    for (auto& v : h) {
        const float f = ((rand() & 0xffff) / 32768.0f - 1.0f) * scale;
        unsigned u;
        std::memcpy(&u, &f, 4);
        v = (unsigned short)(u >> 16);
    }
    hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
}

static void run(bool save, int nb, int B) {
    const int C = 96, heads = 3;
    const size_t nwin = (size_t)nb * B, M = nwin * 49;
    unsigned short *x, *wq, *wp, *y, *qkv = nullptr, *att = nullptr;
    float *bq, *dist, *alpha, *beta, *lse = nullptr;
    hipMalloc(&x, M * C * 2); hipMalloc(&y, M * C * 2); hipMalloc(&wq, 288 * 96 * 2); hipMalloc(&wp, 96 * 96 * 2);
    hipMalloc(&bq, 288 * 4); hipMalloc(&dist, (size_t)nb * 4096 * 4); hipMalloc(&alpha, 169 * heads * 4); hipMalloc(&beta, 169 * heads * 4);
    if (save) { hipMalloc(&qkv, M * 3 * C * 2 + 4096); hipMalloc(&att, M * C * 2); hipMalloc(&lse, nwin * heads * 64 * 4); }
    fill_bf16(x, M * C, 1.0f); fill_bf16(wq, 288 * 96, 0.1f); fill_bf16(wp, 96 * 96, 0.1f);
    hipMemset(bq, 0, 288 * 4); hipMemset(alpha, 0, 169 * heads * 4); hipMemset(beta, 0, 169 * heads * 4);
    {
        std::vector<float> hd((size_t)nb * 4096);
        for (auto& v : hd) v = (rand() & 0xff) / 256.0f;
        hipMemcpy(dist, hd.data(), hd.size() * 4, hipMemcpyHostToDevice);
    }
    auto launch = [&]() {
        return pswin_win_attn_fused_fwd(x, wq, bq, wp, dist, nb, alpha, beta, nullptr, 0, y, qkv, att, lse, (long long)nwin, nb, C, heads, 0.1767767f,
                                        PSWIN_BF16, nullptr);
    };
    if (int rc = launch()) { printf("rc %d\n", rc); return; }
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // ~2 s of back-to-back launches, timing the last 2000
    const int warm = save ? 20000 : 32000, timed = 2000;
    for (int i = 0; i < warm; ++i) launch();
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < timed; ++i) launch();
    hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256][2];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(pswin_fused_clock_probe), sizeof(h));
    std::vector<double> ghz;
    const int grid = nb < 256 ? nb : 256;
    for (int i = 0; i < grid; ++i) if (h[i][1] > 0) ghz.push_back((double)h[i][0] / (double)h[i][1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double us = ms * 1000.0 / timed;
    const double flop = (double)nwin * (2.0 * 49 * 96 * 288 + 3 * 4.0 * 49 * 49 * 32 + 2.0 * 49 * 96 * 96);
    const double issued = (double)nwin * (2.0 * 64 * 96 * 288 + 3 * 4.0 * 64 * 64 * 32 + 2.0 * 64 * 96 * 96);
    const double clk = ghz.empty() ? 0 : ghz[ghz.size() / 2];
    printf("%-9s windows %6zu: %7.2f us/launch  in-kernel clock median %.3f GHz (min %.3f max %.3f, %zu workgroups)  algorithmic %6.1f TFLOP/s = %.1f %% of 2.5 PF"
           "  | issued on 64-row tiles %6.1f TFLOP/s; MFMA peak at the measured clock %.0f TFLOP/s -> algorithmic %.1f %%, issued %.1f %% of it\n",
           save ? "training" : "inference", nwin, us, clk, ghz.empty() ? 0 : ghz.front(), ghz.empty() ? 0 : ghz.back(), ghz.size(), flop / us * 1e-6,
           flop / us * 1e-6 / 2500 * 100, issued / us * 1e-6, 1024 * 1024 * clk * 1e9 * 1e-12 * 1.0, flop / us * 1e-6 / (1024.0 * 1024 * clk * 1e-3) * 100,
           issued / us * 1e-6 / (1024.0 * 1024 * clk * 1e-3) * 100);
}

int main() {
    run(false, 703, 8);
    run(true, 703, 8);
    return 0;
}
