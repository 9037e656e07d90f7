// Ablation probe for pswin_gemm_nt (not part of the library): the same kernel built with PSWIN_NT_PROBE = 0 / 1 / 2, timed standalone.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPSWIN_NT_PROBE=<m> -I include -I panoswintransformerobjectdetection_amd/csrc tools/probe/nt_probe.hip -o /tmp/nt_probe<m>
#include "../../panoswintransformerobjectdetection_amd/csrc/pswin_gemm_nt.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>

static void run(int M, int K, int N, int tile) {
    std::vector<unsigned short> hx((size_t)M * K), hw((size_t)N * K);
    for (auto& v : hx) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) | ((rand() & 1) << 15));
    for (auto& v : hw) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) | ((rand() & 1) << 15));
    unsigned short *x, *w, *y;
    hipMalloc(&x, hx.size() * 2); hipMalloc(&w, hw.size() * 2); hipMalloc(&y, (size_t)M * N * 2);
    hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) pswin_gemm_nt(x, w, nullptr, y, M, K, N, tile, nullptr);
    hipEventRecord(e0, nullptr);
    const int it = 50;
    for (int i = 0; i < it; ++i) pswin_gemm_nt(x, w, nullptr, y, M, K, N, tile, nullptr);
    hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int trow = tile;
    const double us = ms * 1000.0 / it, tiles = (double)((M + trow - 1) / trow) * (N / 192);
    printf("probe %d  M %6d K %5d N %5d tile %3d : %7.2f us  %6.1f TFLOP/s  fill %5.1f TB/s\n", PSWIN_NT_PROBE, M, K, N, tile, us,
           2.0 * M * K * N / us * 1e-6, tiles * (trow + 192) * K * 2 / us * 1e-6);
    hipFree(x); hipFree(w); hipFree(y);
}

static void run_epi(int M, int K, int N, int tile) {
    unsigned short *x, *w, *y, *h;
    float *bias, *part;
    hipMalloc(&x, (size_t)M * K * 2); hipMalloc(&w, (size_t)N * K * 2); hipMalloc(&y, (size_t)M * N * 2); hipMalloc(&h, (size_t)M * N * 2);
    hipMalloc(&bias, N * 4); hipMalloc(&part, (size_t)((M + 63) / 64) * N * 4);
    hipMemset(x, 0x3c, (size_t)M * K * 2); hipMemset(w, 0x3c, (size_t)N * K * 2); hipMemset(y, 0x3c, (size_t)M * N * 2); hipMemset(bias, 0, N * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    const int it = 30;
    for (int mode = 0; mode < 3; ++mode) {
        for (int i = 0; i < it + 3; ++i) {
            if (i == 3) hipEventRecord(e0, nullptr);
            if (mode == 0) pswin_gemm_nt(x, w, bias, y, M, K, N, tile, nullptr);
            if (mode == 1) pswin_gemm_nt_gelu_fwd(x, w, bias, y, h, M, K, N, tile, nullptr);
            if (mode == 2) pswin_gemm_nt_gelu_bwd(x, w, y, bias, h, part, M, K, N, tile, nullptr);
        }
        hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("epi %s M %6d K %5d N %5d tile %3d : %7.2f us\n", mode == 0 ? "plain+bias" : mode == 1 ? "gelu_fwd  " : "gelu_bwd  ", M, K, N, tile, ms * 1000.0 / it);
    }
    hipFree(x); hipFree(w); hipFree(y); hipFree(h); hipFree(bias); hipFree(part);
}

int main() {
    run_epi(16384, 384, 1536, 128); run_epi(65536, 192, 768, 128);
    for (int tile : {128, 64}) {
        run(16384, 384, 1536, tile); run(19600, 384, 1152, tile); run(65536, 192, 768, tile); run(74480, 192, 576, tile);
        run(16384, 1536, 384, tile); run(65536, 768, 192, tile); run(16384, 3072, 1536, tile); run(4096, 3072, 768, tile); run(5880, 768, 2304, tile); run(19600, 384, 384, tile);
    }
    return 0;
}
