// Persistent ring-pipelined bf16 GEMM for the Linear layers of stages 1-3 (gfx950):  Y[M, N] = X[M, K] . W[N, K]^T (+ bias)
//
// Round 3.  The tiled kernel of round 2 (pswin_gemm_nt.hip) gives every 128 x 192 tile its own workgroup: with K = 192 - 768 a tile
// is 3 - 12 k-steps long, so a third of its life is the first loads' latency and the store tail, and two workgroups per CU only
// partly fill each other's bubbles (profiles/r02_gemm_nt_vs_library.txt: loads alone and MFMAs alone each take 3/4 of a launch).
// Here ONE 8-wave workgroup per CU walks its share of the tiles and the k-steps of consecutive tiles form ONE stream through a
// three-stage LDS ring: the operands of steps s + 1 and s + 2 are in flight while step s is multiplied -- across tile boundaries
// too -- and a tile's stores leave while the next tile's steps run.  Waits are counted (vmcnt counts loads and stores in issue
// order: the wait in front of a step allows the younger step's 6 LDS-DMA instructions and, for two steps after an epilogue, its 9
// stores to stay outstanding), the one barrier per step is a raw s_barrier.
//
//   * macro tile 192 x 192 (every N of the model is a multiple of 192), BK = 64; 8 waves as 4 (rows) x 2 (columns), 48 x 96 per
//     wave = 18 accumulator quads, 36 MFMAs per wave and step; ring = 3 x (24 + 24) KB = 144 KB;
//   * operand tiles go global -> LDS by LDS-DMA through buffer resources (rows past M read as zeros), 128-byte rows with the 16-byte
//     chunk XOR-ed by (row & 7) -- applied to the source address and again on the fragment reads (as pswin_gemm_nt.hip);
//   * transposed product (A = weight rows): a lane owns 4 consecutive output columns of one row, column tiles are paired with
//     v_permlane16_swap and rows leave as 16-byte stores;
//   * tiles are dealt so that the workgroups of one XCD (blockIdx % 8 under round-robin placement; speed only) hold consecutive
//     tiles, column tile fastest: they share activation panels in that XCD's L2.
#include <type_traits>
#include <utility>

#include "pswin_common.hpp"

using namespace pswin;

namespace {

constexpr int R_BM = 192, R_BN = 192, R_BK = 64, R_THREADS = 512, R_STAGES = 3;
constexpr int R_A_BYTES = R_BM * R_BK * 2, R_B_BYTES = R_BN * R_BK * 2, R_STAGE = R_A_BYTES + R_B_BYTES;
constexpr int R_RT = 3, R_CT = 6;                      // 16-row / 16-column tiles per wave (48 x 96)
constexpr int R_LOADS = 6;                             // LDS-DMA instructions per wave and step
constexpr int R_STORES = R_RT * R_CT / 2;              // 16-byte stores per wave and tile
using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((address_space(3))) void lds_void;

__device__ inline f32x4 mfma32(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ inline unsigned pk2(float lo, float hi) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2_t)); }
__device__ inline u32x4 pack_row8(f32x4 q0, f32x4 q1) {
    const unsigned a0 = pk2(q0[0], q0[1]), a1 = pk2(q0[2], q0[3]), b0 = pk2(q1[0], q1[1]), b1 = pk2(q1[2], q1[3]);
    const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
    return u32x4{r0[0], r1[0], r0[1], r1[1]};
}

template <int N>
__device__ inline void wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == R_LOADS) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == R_STORES) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    static_assert(N == 0 || N == R_LOADS || N == R_STORES || N == R_LOADS + R_STORES, "");
    static_assert(R_LOADS == 6 && R_STORES == 9, "the immediates above are written out");
}

__global__ __launch_bounds__(R_THREADS, 2) void gemm_nt_ring_kernel(const unsigned short* __restrict__ X, const unsigned short* __restrict__ W,
                                                                    const float* __restrict__ bias, unsigned short* __restrict__ Y, int M, int N,
                                                                    int K, int tiles_n, int ntiles) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int c = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int KT = K / R_BK;

    // this workgroup's tiles: super-step i covers tiles [i G, (i + 1) G); inside it the 8 XCD groups take contiguous chunks
    const int G = gridDim.x, per = G >> 3;             // G is a multiple of 8 (launcher)
    const int slot = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    const int n_my = slot < ntiles ? (ntiles - slot + G - 1) / G : 0;
    if (n_my == 0) return;                              // uniform for the workgroup: no barrier is skipped by a part of it

    // LDS-DMA: waves 0-3 bring the 24 row blocks (8 rows x 128 B) of the activation tile, waves 4-7 those of the weight tile
    const bool mine_a = wave < 4;
    const rsrc_t src = mine_a ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(X), 0, (int)((size_t)M * K * 2), 0x00020000)
                              : __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(W), 0, (int)((size_t)N * K * 2), 0x00020000);
    const int lrow = lane >> 3, lch = (lane & 7) ^ (lrow & 7);
    unsigned lane_off[R_LOADS];                         // byte offset of this lane's 16 bytes relative to the tile's first row, k-step 0
#pragma unroll
    for (int j = 0; j < R_LOADS; ++j) lane_off[j] = (unsigned)(((wave & 3) * R_LOADS + j) * 8 + lrow) * (unsigned)(K * 2) + (unsigned)(lch * 16);
    const int dma_base = (mine_a ? 0 : R_A_BYTES) + (wave & 3) * R_LOADS * 1024;

    int l_it = 0, l_kt = 0, l_stage = 0;                // load cursor: (tile iteration, k-step) of the next slab to request
    unsigned l_row0 = 0;                                // first row of the cursor's tile in this wave's operand (m0 or n0), in bytes * K
    auto cursor_tile = [&](int it) {
        const int t = it * G + slot, tm = t / tiles_n, tn = t - tm * tiles_n;
        l_row0 = (unsigned)((mine_a ? tm * R_BM : tn * R_BN)) * (unsigned)(K * 2);
    };
    cursor_tile(0);
    auto issue_next = [&]() {
        if (l_it < n_my) {
            char* base = smem + l_stage * R_STAGE + dma_base;
            const unsigned off = l_row0 + (unsigned)(l_kt * (R_BK * 2));
#pragma unroll
            for (int j = 0; j < R_LOADS; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (lds_void*)(base + j * 1024), 16, lane_off[j] + off, 0, 0, 0);
            l_stage = l_stage == R_STAGES - 1 ? 0 : l_stage + 1;
            if (++l_kt == KT) {
                l_kt = 0;
                if (++l_it < n_my) cursor_tile(l_it);
            }
        }
    };

    f32x4 acc[R_RT][R_CT];
#pragma unroll
    for (int i = 0; i < R_RT; ++i)
#pragma unroll
        for (int j = 0; j < R_CT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int a_lane = (wm * 48 + c) * 128, b_lane = R_A_BYTES + (wn * 96 + c) * 128;
    int choff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) choff[ks] = ((4 * ks + g) ^ (c & 7)) << 4;
    const int d0 = 8 * (g >> 1) + 16 * (g & 1);
    const rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(Y, 0, (int)((size_t)M * N * 2), 0x00020000);

    // Software pipeline inside a wave: the fragment reads of the NEXT 32-deep half are issued between the MFMAs of the current one
    // (one read per two MFMAs), so that LDS latency runs under the matrix pipe even when the two waves of a SIMD arrive at the
    // per-step barrier together.  Step s: [wait slab s | barrier | request slab s + 2]  A: MFMAs (s - 1, second half) + reads (s, first
    // half)   B: MFMAs (s, first half) + reads (s, second half).  A tile's accumulators are complete after phase A of the step that
    // follows its last k-step: its epilogue sits there.
    auto read_half = [&](auto ks_tag, auto n_tag, const char* sa, const char* sb, u32x4 (&af)[R_RT], u32x4 (&bf)[R_CT]) {
        constexpr int KS = decltype(ks_tag)::value, NN = decltype(n_tag)::value;     // NN 0..8: 3 activation + 6 weight fragments
        if constexpr (NN < R_RT) af[NN] = *reinterpret_cast<const u32x4*>(sa + NN * 2048 + choff[KS]);
        else bf[NN - R_RT] = *reinterpret_cast<const u32x4*>(sb + (NN - R_RT) * 2048 + choff[KS]);
    };
    auto phase = [&](auto ks_tag, auto read_tag, const char* sa, const char* sb, const u32x4 (&ca)[R_RT], const u32x4 (&cb)[R_CT],
                     u32x4 (&na)[R_RT], u32x4 (&nb)[R_CT]) {
        constexpr bool READ = decltype(read_tag)::value;
        auto slot = [&](auto n_tag) {
            constexpr int NN = decltype(n_tag)::value, J = NN / R_RT, I = NN % R_RT;     // MFMA NN of 18: acc[I][J]
            if constexpr (READ && (NN % 2 == 0)) read_half(ks_tag, std::integral_constant<int, NN / 2>{}, sa, sb, na, nb);
            acc[I][J] = mfma32(cb[J], ca[I], acc[I][J]);
            __builtin_amdgcn_sched_barrier(0);
        };
        __builtin_amdgcn_s_setprio(1);
        [&]<int... Ns>(std::integer_sequence<int, Ns...>) { (slot(std::integral_constant<int, Ns>{}), ...); }(std::make_integer_sequence<int, 18>{});
        __builtin_amdgcn_s_setprio(0);
    };
    auto epilogue = [&](int it_done) {
        // acc[i][j][e] = Y[row m0 + 48 wm + 16 i + c][column n0 + 96 wn + 16 j + 4 g + e]
        const int t = it_done * G + slot, tm = t / tiles_n, tn = t - tm * tiles_n;
        const int m0 = tm * R_BM, n0 = tn * R_BN;
#pragma unroll
        for (int i = 0; i < R_RT; ++i) {
            const unsigned row = (unsigned)(m0 + wm * 48 + 16 * i + c);
            const unsigned base = row < (unsigned)M ? row * (unsigned)(N * 2) + (unsigned)((n0 + wn * 96 + d0) * 2) : 0xFFFFFF00u;
#pragma unroll
            for (int jp = 0; jp < R_CT / 2; ++jp) {
                f32x4 q0 = acc[i][2 * jp], q1 = acc[i][2 * jp + 1];
                if (bias) {                             // (an ordinary load here drains the ring once per tile: the biased layers -- qkv --
                    const float* bp = bias + n0 + wn * 96 + 32 * jp + 4 * g;          //  pay about a step for it)
                    q0 += *reinterpret_cast<const f32x4*>(bp);
                    q1 += *reinterpret_cast<const f32x4*>(bp + 16);
                }
                const unsigned off = base == 0xFFFFFF00u ? base : base + 64u * jp;
                __builtin_amdgcn_raw_buffer_store_b128(pack_row8(q0, q1), ys, off, 0, 0);
                acc[i][2 * jp] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc[i][2 * jp + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    using KS0 = std::integral_constant<int, 0>;
    using KS1 = std::integral_constant<int, 1>;

    issue_next();
    issue_next();
    const int total = n_my * KT;
    u32x4 a0[R_RT], b0[R_CT], a1[R_RT], b1[R_CT];       // fragments of the first / second 32-deep half
    int stage = 0, kt = 0, it = 0, since_epi = 3;       // since_epi: steps since the stores of the last epilogue were issued (they are
    for (int s = 0; s < total; ++s) {                   // younger than the operand loads of the two steps that follow)
        const bool has_next = s + 1 < total, stores_young = since_epi < 2;
        if (has_next) {
            if (stores_young) wait_vm<R_LOADS + R_STORES>();
            else wait_vm<R_LOADS>();
        } else {
            if (stores_young) wait_vm<R_STORES>();
            else wait_vm<0>();
        }
        // every read of slab s - 1 has returned (its stage is overwritten by the request below) before any wave passes the barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue_next();                                   // slab s + 2
        const char* sa = smem + stage * R_STAGE + a_lane;
        const char* sb = smem + stage * R_STAGE + b_lane;
        ++since_epi;
        if (s == 0) {
            [&]<int... Ns>(std::integer_sequence<int, Ns...>) { (read_half(KS0{}, std::integral_constant<int, Ns>{}, sa, sb, a0, b0), ...); }(std::make_integer_sequence<int, 9>{});
        } else {
            phase(KS0{}, std::true_type{}, sa, sb, a1, b1, a0, b0);          // MFMAs: second half of slab s - 1 | reads: first half of slab s
            if (kt == 0) {                              // slab s - 1 closed a tile
                epilogue(it - 1);
                since_epi = 0;
            }
        }
        phase(KS1{}, std::true_type{}, sa, sb, a0, b0, a1, b1);              // MFMAs: first half of slab s | reads: its second half
        stage = stage == R_STAGES - 1 ? 0 : stage + 1;
        if (++kt == KT) {
            kt = 0;
            ++it;
        }
    }
    phase(KS0{}, std::false_type{}, smem, smem, a1, b1, a0, b0);             // the last slab's second half
    epilogue(n_my - 1);
}

}  // namespace

extern "C" {

int pswin_gemm_nt_ring_supported(long long M, int K, int N) {
    return M >= 192 && (M + 192) * (long long)(K > N ? K : N) * 2 < 0xFFFFFF00ll && K >= 192 && K % 64 == 0 && N >= 192 && N % 192 == 0;
}

int pswin_gemm_nt_ring(const void* x, const void* w, const float* bias, void* y, long long M, int K, int N, int max_wgs, void* stream) {
    PSWIN_CHECK_ARG(x && w && y && pswin_gemm_nt_ring_supported(M, K, N));
    PSWIN_CHECK_ARG(aligned16(x) && aligned16(w) && aligned16(y) && aligned16(bias));
    constexpr size_t lds = (size_t)R_STAGES * R_STAGE;
    static std::atomic<unsigned long long> configured{0};
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&gemm_nt_ring_kernel), lds, configured)) return rc;
    const int m = (int)M, tiles_m = (m + R_BM - 1) / R_BM, tiles_n = N / R_BN, ntiles = tiles_m * tiles_n;
    int grid = max_wgs > 0 ? max_wgs : 256;             // one persistent workgroup per CU
    if (grid > ntiles) grid = ntiles;
    grid = (grid + 7) / 8 * 8;
    hipLaunchKernelGGL(gemm_nt_ring_kernel, dim3(grid), dim3(R_THREADS), lds, (hipStream_t)stream, reinterpret_cast<const unsigned short*>(x),
                       reinterpret_cast<const unsigned short*>(w), bias, reinterpret_cast<unsigned short*>(y), m, N, K, tiles_n, ntiles);
    PSWIN_LAUNCH_RET();
}

}  // extern "C"
