// In-kernel interval probe for pswin_attn_bwd (not part of the library): the shipped kernel built with -DPSWIN_ATTN_STAMPS, which
// accumulates s_memtime intervals per wave between the phases of the batch loop, at the PanoSwin-T stage shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DPSWIN_ATTN_STAMPS -I include -I panoswintransformerobjectdetection_amd/csrc \
//         tools/probe/attn_probe.hip -o tools/probe/bin/attn_probe
#include "../../panoswintransformerobjectdetection_amd/csrc/pswin_attn.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

static void run(const char* name, int nb, int B, int heads, int chunks) {
    const int C = heads * 32, nwin = nb * B;
    const size_t M = (size_t)nwin * 49;
    std::vector<unsigned short> h(M * 3 * C);
    for (auto& v : h) v = (unsigned short)((0x3c00 + (rand() & 0x1ff)) | ((rand() & 1) << 15));
    unsigned short *qkv, *dout, *dqkv;
    float *dist, *alpha, *beta, *lse;
    unsigned long long* st;
    hipMalloc(&qkv, M * 3 * C * 2); hipMalloc(&dout, M * C * 2); hipMalloc(&dqkv, M * 3 * C * 2);
    hipMalloc(&dist, (size_t)nb * 4096 * 4); hipMalloc(&alpha, 169 * heads * 4); hipMalloc(&beta, 169 * heads * 4);
    hipMalloc(&lse, (size_t)nwin * heads * 64 * 4);
    const int items = chunks * nb * heads;
    hipMalloc(&st, (size_t)items * 2 * 8 * 8);
    hipMemcpy(qkv, h.data(), M * 3 * C * 2, hipMemcpyHostToDevice);
    hipMemcpy(dout, h.data(), M * C * 2, hipMemcpyHostToDevice);
    hipMemset(dist, 0, (size_t)nb * 4096 * 4); hipMemset(alpha, 0, 169 * heads * 4); hipMemset(beta, 0, 169 * heads * 4);
    std::vector<float> hl((size_t)nwin * heads * 64, 8.0f);
    hipMemcpy(lse, hl.data(), hl.size() * 4, hipMemcpyHostToDevice);
    pswin_attn_debug_stamps(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int i = 0; i < 6; ++i) {
        if (i == 1) hipEventRecord(e0, nullptr);
        int rc = pswin_attn_bwd(qkv, qkv + C, qkv + 2 * C, 3 * C, dist, nb, alpha, beta, nullptr, 0, dout, C, lse, dqkv, dqkv + C, dqkv + 2 * C, 3 * C,
                                nullptr, chunks, nwin, nb, heads, 0.1767767f, PSWIN_BF16, nullptr);
        if (rc) { printf("rc %d\n", rc); return; }
    }
    hipEventRecord(e1, nullptr); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hs((size_t)items * 2 * 8);
    hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
    double sum[8] = {0};
    for (size_t i = 0; i < hs.size(); ++i) sum[i % 8] += (double)hs[i];
    const double nw = (double)items * 2, reps = (double)B / chunks;
    printf("%-8s nb %4d heads %2d chunks %d items %5d: %7.1f us/launch | per wave (s_memtime ticks): preamble %7.0f | per image: scores+softmax %6.0f"
           "  dV/dK partial %6.0f  barrier %6.0f  finish dV/dK %6.0f  dQ %6.0f  barrier %6.0f | total per wave %8.0f\n",
           name, nb, heads, chunks, items, ms * 1000 / 5, sum[0] / nw, sum[1] / nw / reps, sum[2] / nw / reps, sum[3] / nw / reps, sum[4] / nw / reps,
           sum[5] / nw / reps, sum[6] / nw / reps, (sum[0] + sum[1] + sum[2] + sum[3] + sum[4] + sum[5] + sum[6]) / nw);
    hipFree(qkv); hipFree(dout); hipFree(dqkv); hipFree(dist); hipFree(alpha); hipFree(beta); hipFree(lse); hipFree(st);
}

int main() {
    for (int ch : {1, 2, 4}) run("stage0", 703, 8, 3, ch);
    for (int ch : {1, 2}) run("stage1", 190, 8, 6, ch);
    for (int ch : {1, 2, 4}) run("stage2", 50, 8, 12, ch);
    for (int ch : {1, 2, 4, 8}) run("stage3", 15, 8, 24, ch);
    return 0;
}
