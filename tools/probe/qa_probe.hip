// Where a wave of the fused qkv + attention kernel spends its time (not part of the library): pswin_qkvattn.hip built with
// -DPSWIN_QA_PROBE stamps s_memrealtime (100 MHz) at the phase boundaries of every wave.  Launched ~1 s back to back on random data,
// then the stamps of the last launch are averaged over workgroups and waves and printed per phase, next to the launch time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -DPSWIN_QA_PROBE -I include -I panoswintransformerobjectdetection_amd/csrc \
//         tools/probe/qa_probe.hip -o tools/probe/bin/qa_probe
#include "../../panoswintransformerobjectdetection_amd/csrc/pswin_qkvattn.hip"
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

static unsigned long long h[256][QWAVES][QA_SLOTS];

static void run(int C, int nb, int B, bool save) {
    const int heads = C / 32;
    const size_t nwin = (size_t)nb * B, M = nwin * 49;
    unsigned short *x, *wq, *y, *qkv = nullptr;
    float *bq, *dist, *alpha, *beta, *lse = nullptr;
    hipMalloc(&x, M * C * 2); hipMalloc(&y, M * C * 2); hipMalloc(&wq, (size_t)3 * C * C * 2);
    hipMalloc(&bq, 3 * C * 4); hipMalloc(&dist, (size_t)nb * 4096 * 4); hipMalloc(&alpha, 169 * heads * 4); hipMalloc(&beta, 169 * heads * 4);
    if (save) { hipMalloc(&qkv, M * 3 * C * 2 + 4096); hipMalloc(&lse, nwin * heads * 64 * 4); }
    fill_bf16(x, M * C, 1.0f); fill_bf16(wq, (size_t)3 * C * C, 0.05f);
    hipMemset(bq, 0, 3 * C * 4); hipMemset(alpha, 0, 169 * heads * 4); hipMemset(beta, 0, 169 * heads * 4);
    {
        std::vector<float> hd((size_t)nb * 4096);
        for (auto& v : hd) v = (rand() & 0xff) / 256.0f;
        hipMemcpy(dist, hd.data(), hd.size() * 4, hipMemcpyHostToDevice);
    }
    auto launch = [&]() {
        return pswin_qkv_attn_fused_fwd(x, wq, bq, dist, nb, alpha, beta, nullptr, 0, y, qkv, lse, (long long)nwin, nb, C, heads, 0.1767767f, PSWIN_BF16, nullptr);
    };
    if (int rc = launch()) { printf("rc %d\n", rc); return; }
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int warm = 15000, timed = 2000;
    for (int i = 0; i < warm; ++i) launch();
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < timed; ++i) launch();
    hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpyFromSymbol(h, HIP_SYMBOL(pswin_qa_probe), sizeof(h));
    const int grid = (long long)nb * heads < 256 ? nb * heads : 256 / heads * heads;
    const double flop = (double)nwin * (2.0 * 49 * C * 3 * C + heads * 4.0 * 49 * 49 * 32);
    printf("C %d %s: %d bias windows x %d images, grid %d, %.2f us/launch, algorithmic %.1f TFLOP/s = %.1f %% of 2.5 PF\n", C, save ? "training" : "inference", nb, B, grid,
           ms * 1000.0 / timed, flop / (ms * 1000.0 / timed) * 1e-6, flop / (ms * 1000.0 / timed) * 1e-6 / 25);
    // phases, averaged over every wave that ran them (ticks of 10 ns -> us)
    auto avg = [&](int k0, int k1, int wg_lo, int wg_hi, int* cnt = nullptr) {
        double s = 0; int n = 0;
        for (int b = wg_lo; b < wg_hi; ++b)
            for (int w = 0; w < QWAVES; ++w)
                if (h[b][w][k0] && h[b][w][k1]) { s += (double)(h[b][w][k1] - h[b][w][k0]) * 0.01; ++n; }
        if (cnt) *cnt = n;
        return n ? s / n : 0.0;
    };
    int n;
    printf("  all workgroups: entry -> weights staged %.2f | -> barrier %.2f | first bias + ring prologue + barrier %.2f us\n", avg(0, 1, 0, grid), avg(1, 2, 0, grid), avg(2, 3, 0, grid));
    for (int it = 0; it < 6; ++it) {
        const int k = 4 + 6 * it;
        const double t_k = avg(k, k + 1, 0, grid, &n);
        if (!n) break;
        printf("  item %d (%4d waves): qkv steps %.2f | pack / saves %.2f | scores-softmax-PV %.2f | next bias %.2f | barrier wait %.2f | item total %.2f us\n", it, n, t_k,
               avg(k + 1, k + 2, 0, grid), avg(k + 2, k + 3, 0, grid), avg(k + 3, k + 4, 0, grid), avg(k + 4, k + 5, 0, grid), avg(k, k + 5, 0, grid));
    }
    // whole-kernel span per workgroup: from the earliest entry stamp of the grid to each workgroup's last stamp
    unsigned long long t0 = ~0ull, t1 = 0, tmax_entry = 0;
    for (int b = 0; b < grid; ++b)
        for (int w = 0; w < QWAVES; ++w) {
            if (h[b][w][0]) { t0 = std::min(t0, h[b][w][0]); tmax_entry = std::max(tmax_entry, h[b][w][0]); }
            for (int k = 0; k < QA_SLOTS; ++k) t1 = std::max(t1, h[b][w][k]);
        }
    printf("  first entry -> last entry %.2f us, first entry -> last stamp %.2f us\n", (double)(tmax_entry - t0) * 0.01, (double)(t1 - t0) * 0.01);
    hipFree(x); hipFree(y); hipFree(wq); hipFree(bq); hipFree(dist); hipFree(alpha); hipFree(beta);
    if (save) { hipFree(qkv); hipFree(lse); }
}

int main() {
    run(384, 50, 8, false);
    run(384, 50, 8, true);
    run(192, 190, 8, false);
    run(192, 190, 8, true);
    return 0;
}
