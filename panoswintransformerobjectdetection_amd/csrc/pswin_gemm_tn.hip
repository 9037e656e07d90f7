// Weight-gradient GEMM of the Linear layers of stages 1-3 (gfx950):  dW[N, K] = dY[M, N]^T . X[M, K]
//
// The contraction runs over the M = 4k-75k token rows, i.e. over the STRIDED index of both row-major operands ("TN").
// PyTorch-ROCm serves it as a batched split-K library GEMM at 0.33 of its floor (2.05 ms of a 12.3 ms PanoSwin-T step,
// profiles/r02).  Here: 64-row slabs of dY and X go global -> LDS by LDS-DMA (buffer_load ... lds: rows past M read as
// zeros through the buffer range check), and BOTH MFMA operands are read from the row-major LDS panels TRANSPOSED with
// ds_read_b64_tr_b16 (two 8-byte reads = the 8 consecutive rows m of one column that a lane contributes to a 32-deep step).
//
// Design (CDNA4):
//   * LDS panels of [64 rows m][64 columns] bf16 (128-byte rows); the 32-byte block index of a row is XOR-ed with
//     ((m >> 1) & 1) | (((m >> 3) & 1) << 1): the 8 (row, block) pieces a half-wave's transposed read touches then fall on 8
//     different 32-byte slots of the 256-byte bank row (conflict-free); LDS-DMA writes lane-linear, so the permutation is
//     applied to the source column of each lane;
//   * one 8-wave workgroup per CU owns one output tile x one row range; the slabs travel through a three-stage ring with
//     counted waits (details at the kernel below; the two-stage predecessor of round 2 was removed in round 4);
//   * the M rows are split over `splits` workgroups per output tile (the tiles alone are 1-64 workgroups); each split writes
//     its partial tile [split][N][K] once (the product is computed as X^T . dY so that a lane owns 4 consecutive k of one
//     row n) and the caller sums the splits in its fixed-order grouped reduction (pswin_reduce_jobs) straight into the flat
//     gradient buffer -- no atomics, bitwise reproducible.
#include <type_traits>

#include "pswin_common.hpp"

using namespace pswin;

namespace {

constexpr int MSTEP = 64, PANEL = 64 * 64 * 2;          // one [64][64] bf16 panel: 8 KB
using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef __attribute__((address_space(3))) void lds_void;

__device__ inline f32x4 mfma32(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// ---------------------------------------------------------------------------------------------------------------------
// Round 3: the same product with a THREE-stage LDS ring.  The contraction of a weight gradient is long (256 - 2,400 rows per
// split = 4 - 37 slabs), so unlike the forward GEMMs (3 - 12 k-steps per tile) a deep pipeline reaches steady state: the slabs of
// steps t + 1 and t + 2 are in flight while step t is multiplied, the wait in front of a step is a COUNTED vmcnt that leaves the
// younger slab's LDS-DMA outstanding, and the one barrier per step is a raw s_barrier (a __syncthreads() would drain vmcnt to 0:
// cdna_hip_programming.md, "Pipelining across barriers").  One workgroup of 8 waves per CU (2 waves per SIMD), macro tile
// 192 (k) x 192 (n) -- every N and K of the model is a multiple of 192, so one kernel serves every layer, the stage-1 qkv / proj
// pair included -- waves as 2 (k) x 4 (n), 18 accumulator quads and 36 MFMAs per wave and slab; ring = 3 x 6 panels = 144 KB.
// OUT_BF16: the split's partial tile is rounded to bf16 (what the library's batched GEMM writes for its row chunks) -- half the
// partial-slab traffic of the f32 form, same fixed-order f32 sum of the splits afterwards.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int RING_THREADS = 512;
// Geometry of one instantiation: the 8 waves form a WK (k) x WN (n) grid of 96 x 48 wave tiles -> macro tile 96 WK x 48 WN, whose
// columns come from PX = ceil(96 WK / 64) panels of X and PD = ceil(48 WN / 64) panels of dY per 64-row slab.
//   <2, 4, 3 stages>  192 x 192: every Linear of stages 1-3 (N, K multiples of 192), 48 KB per slab, three slabs in LDS
//   <1, 8, 2 stages>   96 x 384: stage 0 with K = 96 (qkv N = 288, proj N = 96, fc1 N = 384): the WHOLE gradient in one tile, pure
//   <4, 2, 2 stages>  384 x  96: stage-0 fc2 (K = 384, N = 96)                              streaming of dY and X, 64 KB per slab
// A panel whose 64 columns run past the operand's row (K = 96: the second X panel; N = 288 / 96) reads the next row's first columns
// there -- finite values that only ever meet accumulator columns the epilogue does not store; panels entirely past N re-read the last
// needed one (cache hits) so that every wave issues the same number of LDS-DMA instructions per slab (the counted waits need that).
template <int WK, int WN, int STAGES>
struct RingGeom {
    static constexpr int PX = (96 * WK + 63) / 64, PD = (48 * WN + 63) / 64, PANELS = PX + PD, STAGE_BYTES = PANELS * PANEL;
    static constexpr int LOADS = PANELS;                // row blocks per wave and slab: PANELS * 8 blocks over 8 waves
    static constexpr int XWAVES = PX;                   // waves 0 .. PX-1 bring the X panels (8 blocks = one panel each... see issue)
    static_assert(WK * WN == 8, "eight waves");
    static_assert((size_t)STAGES * STAGE_BYTES <= 160 * 1024, "LDS");
};

// The transposed LDS reads of the ring kernel are inline asm: hipcc (ROCm 7.2) treats the ds_read_tr16 builtin as possibly aliasing
// every LDS-DMA in flight and puts `s_waitcnt vmcnt(0)` in front of the first read of each step (seen in the ISA with either LDS-DMA
// builtin) -- which drains the ring.  An asm read is invisible to that pass; its completion is counted by hand: every destination is
// an in-out operand of the wait statement that follows, so no use (or copy) of it can be scheduled in front of the wait
// (cdna_hip_programming.md 5.7, form (ii)).
template <int OFF>
__device__ inline void ds_tr(u32x2& dst, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF) : "memory");
}
struct Frag {
    u32x2 lo, hi;                                       // rows m .. m+3 / m+4 .. m+7 of the lane's column
};
__device__ inline bf16x8 frag_bf16(const Frag& f) { return __builtin_bit_cast(bf16x8, u32x4{f.lo[0], f.lo[1], f.hi[0], f.hi[1]}); }
// wait for every LDS read issued so far; the 9 fragments ride along as in-out operands (18 registers pairs per statement)
__device__ inline void wait_frags(Frag (&a)[6], Frag (&b)[3]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0].lo), "+v"(a[0].hi), "+v"(a[1].lo), "+v"(a[1].hi), "+v"(a[2].lo), "+v"(a[2].hi), "+v"(a[3].lo), "+v"(a[3].hi),
                   "+v"(a[4].lo), "+v"(a[4].hi), "+v"(a[5].lo), "+v"(a[5].hi), "+v"(b[0].lo), "+v"(b[0].hi), "+v"(b[1].lo), "+v"(b[1].hi),
                   "+v"(b[2].lo), "+v"(b[2].hi)
                 :
                 : "memory");
}

// Round 4: the kernel takes a TABLE of independent products (job = one Linear's dW; the table travels in the kernel arguments, as
// pswin_reduce_jobs' does).  Why: a weight gradient is not needed before the backward pass ends, so the host can hold the ~50 of them
// back and issue them together (ops.py: deferred weight gradients).  One launch of many products needs no 256-way row split per
// product to fill the chip -- every workgroup then contracts >= 2,048 rows instead of ~1,000 (at batch 2: instead of ~250), which
// halves (batch 2: removes most of) the partial-slab traffic and the per-launch ramps: 51 launches of 15-30 us become three.
struct TnJob {
    const unsigned short* dy;
    const unsigned short* x;
    void* part;
    float* db;
    int M, N, K, tiles_k, tiles_n, rows_per_split, zero_lo, zero_hi;
    int first_wg, n_wg, flags, xmap;                   // flags = out_bf16 | splits << 1;  xmap = g | gn << 8 | bk << 16 | bn << 24 (below)
};
constexpr int TN_JOBS_MAX = 48;                        // 48 x 80 B: the kernel-argument segment holds 4 KB
struct TnBatch {
    TnJob job[TN_JOBS_MAX];
    int n;
};

template <int WK, int WN, int STAGES>
__global__ __launch_bounds__(RING_THREADS, 2) void gemm_tn_ring_kernel(const TnBatch batch) {
    using G = RingGeom<WK, WN, STAGES>;
    // this workgroup's job: first_wg ascending (multiples of 8, so that a job's workgroups keep the round-robin XCD pattern of a launch
    // of its own); workgroups past a job's count are padding and leave at once
    int jlo = 0, jhi = batch.n - 1;
    while (jlo < jhi) {
        const int mid = (jlo + jhi + 1) >> 1;
        if (batch.job[mid].first_wg <= (int)blockIdx.x) jlo = mid;
        else jhi = mid - 1;
    }
    const TnJob& job = batch.job[jlo];
    const int jbid = (int)blockIdx.x - job.first_wg;
    if (jbid >= job.n_wg) return;
    const unsigned short* __restrict__ DY = job.dy;
    const unsigned short* __restrict__ X = job.x;
    void* __restrict__ P = job.part;
    float* __restrict__ DB = job.db;
    const int M = job.M, N = job.N, K = job.K, tiles_k = job.tiles_k, tiles_n = job.tiles_n, rows_per_split = job.rows_per_split;
    const int zero_lo = job.zero_lo, zero_hi = job.zero_hi;
    const bool out_bf16 = (job.flags & 1) != 0;
    constexpr int IA = 6, JB = 3;                      // 16-wide tiles per wave along k (96 columns) / n (48 columns)
    constexpr int RING_STAGE_BYTES = G::STAGE_BYTES, LOADS = G::LOADS;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int c = lane & 15, g = lane >> 4;
    const int wk = wave / WN, wn = wave - wk * WN;

    // Workgroup -> (row split, tile), XCD-aware (workgroup b of a launch runs on XCD b % 8, each XCD with an L2 of its own; measured per
    // product with FETCH_SIZE, profiles/r04_tn_ring_fetch_per_product.txt: the tiles of a row split that run on ONE XCD fetch its rows once,
    // every further XCD that holds tiles of the split fetches them again).  A split is given to g = 1, 2, 4 or 8 XCDs (1 when there are at
    // least 8 splits: XCD x takes the splits x, x + 8, ...); with g > 1 the split's tile grid is cut into gk x gn blocks of bk x bn tiles,
    // one block per XCD, the cut chosen on the host for the fewest operand columns fetched.  Slots past a block's edge or past the last
    // split leave at once.
    int split, tk, tn;
    {
        const int xm = job.xmap, g = xm & 255, gn = (xm >> 8) & 255, bk = (xm >> 16) & 255, bn = (xm >> 24) & 255;
        const int xcd = jbid & 7, loc = jbid >> 3, lane = xcd / g, blk = xcd - lane * g, ik = blk / gn, in = blk - ik * gn;
        const int per = bk * bn, round = loc / per, within = loc - round * per, wk_ = within / bn;
        split = round * (8 / g) + lane;
        tk = ik * bk + wk_;
        tn = in * bn + (within - wk_ * bn);
        if (split >= (job.flags >> 1) || tk >= tiles_k || tn >= tiles_n) return;
    }
    const int k0 = tk * (96 * WK), n0 = tn * (48 * WN);
    const int m_begin = split * rows_per_split;
    int steps = rows_per_split / MSTEP;
    {
        const int left = (M - m_begin + MSTEP - 1) / MSTEP;           // slabs that hold at least one real row
        steps = left < steps ? (left > 0 ? left : 0) : steps;
    }

    // LDS-DMA: PANELS x 8 row blocks (8 rows x 128 B) per slab = PANELS blocks per wave: wave w brings blocks w * PANELS .. of the
    // list [X panels | dY panels]; PANELS is 6 or 8 and PX * 8 a multiple of it, so a wave serves one operand; rows past M read as
    // zeros (buffer range check)
    static_assert((G::PX * 8) % LOADS == 0, "a wave's blocks must not straddle the two operands");
    const bool mine_x = wave * LOADS < G::PX * 8;
    const rsrc_t src = mine_x ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(X), 0, (int)((size_t)M * K * 2), 0x00020000)
                              : __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(DY), 0, (int)((size_t)M * N * 2), 0x00020000);
    const unsigned ld2 = (unsigned)((mine_x ? K : N) * 2);
    const int c0 = mine_x ? k0 : n0;
    const int lr = lane >> 3, pch = lane & 7;
    unsigned goff[8];                                   // (LOADS <= 8 used; a dependent array bound captured by the lambdas below made hipcc's host pass drop the kernel stub)
    const int last_panel = ((mine_x ? K : N) - c0 + 63) / 64 - 1;     // the last panel that holds a column of this operand's tile
#pragma unroll
    for (int j = 0; j < LOADS; ++j) {
        const int blk_all = wave * LOADS + j - (mine_x ? 0 : G::PX * 8), blk = blk_all & 7;
        int panel = blk_all >> 3;
        panel = panel > last_panel ? (last_panel > 0 ? last_panel : 0) : panel;
        const int row = 8 * blk + lr;
        const int f = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
        const int col = (((pch >> 1) ^ f) << 4) + ((pch & 1) << 3);
        goff[j] = (unsigned)row * ld2 + (unsigned)((c0 + 64 * panel + col) * 2);
    }
    auto issue = [&](int step, int stage) {
        char* base = smem + stage * RING_STAGE_BYTES + wave * LOADS * 1024;
        const unsigned m = (unsigned)(m_begin + step * MSTEP) * ld2;
#pragma unroll
        for (int j = 0; j < LOADS; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (lds_void*)(base + j * 1024), 16, goff[j] + m, 0, 0, 0);
    };

    f32x4 acc[IA][JB];
#pragma unroll
    for (int i = 0; i < IA; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Bias gradient (DB != null): the column sums of dY over this split's rows are one more product, ones^T . dY, on the dY fragments
    // the wave already holds -- 3 MFMAs per 32-row half on top of 18, in the waves of the first k block of the first k tile only.
    // It replaces a separate pass over dY (pswin_colsum: 12 launches per step for the qkv biases).
    const bool sum_db = DB != nullptr && tk == 0 && wk == 0;
    f32x4 accs[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) accs[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});

    // transposed reads: lane (c = 4 q + p, g) addresses row (32 ks + 8 g + q), 8-byte piece p of the 16-column block cb of a panel;
    // block swizzle f = ((q >> 1) & 1) | ((g & 1) << 1) is a lane constant.  One base register per 16-column block position cb and
    // immediate offsets for panel, ks and the second row quad.
    const int q = c >> 2, p8 = c & 3;
    const int fl = ((q >> 1) & 1) | ((g & 1) << 1);
    const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem;
    const unsigned lane_row = lds0 + (unsigned)((8 * g + q) * 128 + p8 * 8);
    unsigned xoff[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) xoff[cb] = ((unsigned)(cb ^ fl) << 5) + lane_row;
    // column block of fragment i (k side) / j (n side) of this wave: col0 = 96 wk + 16 i -> panel col0 >> 6, cb = (col0 >> 4) & 3
    // wk, wn are wave-uniform but not compile-time: the panel part goes into the base register, the rest are immediates
    unsigned abase[IA], bbase[JB];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        const int col0 = wk * 96 + 16 * i;
        abase[i] = xoff[(col0 >> 4) & 3] + (unsigned)((col0 >> 6) * PANEL);
    }
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const int col0 = wn * 48 + 16 * j;
        bbase[j] = xoff[(col0 >> 4) & 3] + (unsigned)((col0 >> 6) * PANEL + G::PX * PANEL);
    }
    // One MFMA and one transposed read per slot: read n of a set is half (n & 1) of fragment n >> 1 (0..5: k side, 6..8: n side), MFMA
    // n is acc[n % 6][n / 6].  The reads of the NEXT 32-row half are issued between the MFMAs of the current one, so their LDS latency
    // runs under the matrix pipe inside ONE wave -- the two waves of a SIMD reach the per-slab barrier together and would otherwise
    // both read, then both multiply (measured: 2,500 cycles per slab for 1,152 cycles of MFMAs with block-wise reads).
    auto read_n = [&](auto ks_tag, auto n_tag, unsigned stage_off, Frag (&a)[6], Frag (&b)[3]) {
        constexpr int KS = decltype(ks_tag)::value, NN = decltype(n_tag)::value, F = NN >> 1, OFF = KS * 4096 + (NN & 1) * 512;
        if constexpr (F < 6) {
            if constexpr (NN & 1) ds_tr<OFF>(a[F].hi, abase[F] + stage_off);
            else ds_tr<OFF>(a[F].lo, abase[F] + stage_off);
        } else {
            if constexpr (NN & 1) ds_tr<OFF>(b[F - 6].hi, bbase[F - 6] + stage_off);
            else ds_tr<OFF>(b[F - 6].lo, bbase[F - 6] + stage_off);
        }
    };
    auto mma_n = [&](auto n_tag, const bf16x8 (&af)[6], const bf16x8 (&bfr)[3]) {
        constexpr int NN = decltype(n_tag)::value, I = NN % 6, J = NN / 6;
        acc[I][J] = mfma32(af[I], bfr[J], acc[I][J]);
    };
    // phase: MFMAs on the fragments of `cur`, reads of half KS of the slab at stage_off into `nxt` (READ false: MFMAs only)
    auto phase = [&](auto ks_tag, auto read_tag, unsigned stage_off, Frag (&cur_a)[6], Frag (&cur_b)[3], Frag (&nxt_a)[6], Frag (&nxt_b)[3]) {
        constexpr bool READ = decltype(read_tag)::value;
        bf16x8 af[IA], bfr[JB];
#pragma unroll
        for (int i = 0; i < IA; ++i) af[i] = frag_bf16(cur_a[i]);
#pragma unroll
        for (int j = 0; j < JB; ++j) bfr[j] = frag_bf16(cur_b[j]);
        auto slot = [&](auto n_tag) {
            if constexpr (READ) read_n(ks_tag, n_tag, stage_off, nxt_a, nxt_b);
            mma_n(n_tag, af, bfr);
            __builtin_amdgcn_sched_barrier(0);
        };
        [&]<int... Ns>(std::integer_sequence<int, Ns...>) { (slot(std::integral_constant<int, Ns>{}), ...); }(std::make_integer_sequence<int, 18>{});
        if (sum_db) {
#pragma unroll
            for (int j = 0; j < JB; ++j) accs[j] = mfma32(ones, bfr[j], accs[j]);
        }
    };
    auto read_all = [&](auto ks_tag, unsigned stage_off, Frag (&a)[6], Frag (&b)[3]) {
        [&]<int... Ns>(std::integer_sequence<int, Ns...>) { (read_n(ks_tag, std::integral_constant<int, Ns>{}, stage_off, a, b), ...); }(std::make_integer_sequence<int, 18>{});
    };
    using KS0 = std::integral_constant<int, 0>;
    using KS1 = std::integral_constant<int, 1>;

    static_assert(STAGES == 2 || STAGES == 3, "");
    static_assert(STAGES == 2 || LOADS == 6, "the counted wait below is written out for 6 loads per slab");
    if (steps > 0) issue(0, 0);
    if (STAGES == 3 && steps > 1) issue(1, 1);
    Frag a0[6], b0[3], a1[6], b1[3];
    int stage = 0;
    for (int st = 0; st < steps; ++st) {
        // slab st has landed for this wave once at most the younger slabs' LDS-DMA instructions (three stages: the 6 of slab st + 1;
        // two stages: none) are outstanding; the barrier then says so for every wave, and that every wave has the previous slab in
        // registers (the wait on a1 / b1 below), so the stage the next issue overwrites is free
        if (STAGES == 3 && st + 1 < steps) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (st > 0) wait_frags(a1, b1);                  // second half of slab st - 1: read during its first half's MFMAs
        __builtin_amdgcn_s_barrier();
        const unsigned stage_off = (unsigned)(stage * RING_STAGE_BYTES);
        if (st + STAGES - 1 < steps) issue(st + STAGES - 1, stage == 0 ? STAGES - 1 : stage - 1);       // (stage + STAGES - 1) % STAGES
        if (st == 0) {
            read_all(KS0{}, stage_off, a0, b0);
        } else {
            phase(KS0{}, std::true_type{}, stage_off, a1, b1, a0, b0);       // MFMAs of slab st - 1, second half | reads of slab st, first half
        }
        wait_frags(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        phase(KS1{}, std::true_type{}, stage_off, a0, b0, a1, b1);           // MFMAs of slab st, first half | reads of its second half
        stage = stage == STAGES - 1 ? 0 : stage + 1;
    }
    if (steps > 0) {
        wait_frags(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        phase(KS0{}, std::false_type{}, 0u, a1, b1, a0, b0);                 // the last slab's second half
    }

    // accs[j][e] = sum over the split's rows of dY[row][n = n0 + 48 wn + 16 j + c], the same in every element and lane group
    if (sum_db && g == 0) {
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const int n = n0 + wn * 48 + 16 * j + c;
            if (n < N) DB[(size_t)split * N + n] = (n >= zero_lo && n < zero_hi) ? 0.f : accs[j][0];
        }
    }
    // acc[i][j][e] = partial dW[n = n0 + 48 wn + 16 j + c][k = k0 + 96 wk + 16 i + 4 g + e]
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const int n = n0 + wn * 48 + 16 * j + c;
        if (n >= N) continue;                           // (tiles wider than the operand: stage 0)
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const int k = k0 + wk * 96 + 16 * i + 4 * g;
            if (k >= K) continue;
            const size_t o = (size_t)split * N * K + (size_t)n * K + k;
            if (out_bf16) {
                *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(P) + o) = u32x2{pack2_bf16(acc[i][j][0], acc[i][j][1]), pack2_bf16(acc[i][j][2], acc[i][j][3])};
            } else {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(P) + o) = acc[i][j];
            }
        }
    }
}

// geometry for a shape: 0 = <2, 4> (192 x 192 tiles), 1 = <1, 8> (K = 96, N <= 384), 2 = <4, 2> (K = 384, N = 96), -1 = none
inline int ring_geom(int N, int K) {
    if (N >= 192 && K >= 192 && N % 192 == 0 && K % 192 == 0) return 0;
    if (K == 96 && N >= 48 && N <= 384 && N % 16 == 0) return 1;
    if (K == 384 && N == 96) return 2;
    return -1;
}

template <int WK, int WN, int STAGES>
int launch_tn_batch(const TnBatch& b, int wgs, hipStream_t st) {
    using G = RingGeom<WK, WN, STAGES>;
    constexpr size_t lds = (size_t)STAGES * G::STAGE_BYTES;
    static std::atomic<unsigned long long> configured{0};
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&gemm_tn_ring_kernel<WK, WN, STAGES>), lds, configured)) return rc;
    hipLaunchKernelGGL((gemm_tn_ring_kernel<WK, WN, STAGES>), dim3(wgs), dim3(RING_THREADS), lds, st, b);
    PSWIN_LAUNCH_RET();
}

inline bool tn_job_ok(const pswin_tn_job& q) {
    return q.dy && q.x && q.partial && q.M >= 64 && (q.M + 64) * (long long)(q.K > q.N ? q.K : q.N) * 2 < 0xFFFFFF00ll && ring_geom(q.N, q.K) >= 0 &&
           q.splits >= 1 && q.splits <= q.M / 64 && valid_dtype(q.partial_dtype) && aligned16(q.dy) && aligned16(q.x) && aligned16(q.partial) &&
           q.zero_lo >= 0 && q.zero_hi >= q.zero_lo && q.zero_hi <= q.N;
}

// every job of `geom` in the caller's order, <= TN_JOBS_MAX per launch; the caller lists the longest row ranges first (the hardware
// deals workgroups in order: longest-first keeps the tail of the launch short)
int launch_tn_jobs(const pswin_tn_job* jobs, int n_jobs, int geom, hipStream_t st) {
    TnBatch b;
    b.n = 0;
    long long wgs = 0;
    auto flush = [&]() -> int {
        if (b.n == 0) return PSWIN_OK;
        int rc;
        if (geom == 0) rc = launch_tn_batch<2, 4, 3>(b, (int)wgs, st);
        else if (geom == 1) rc = launch_tn_batch<1, 8, 2>(b, (int)wgs, st);
        else rc = launch_tn_batch<4, 2, 2>(b, (int)wgs, st);
        b.n = 0;
        wgs = 0;
        return rc;
    };
    for (int i = 0; i < n_jobs; ++i) {
        const pswin_tn_job& q = jobs[i];
        if (ring_geom(q.N, q.K) != geom) continue;
        const int tk = geom == 0 ? 192 : (geom == 1 ? 96 : 384), tn = geom == 0 ? 192 : (geom == 1 ? 384 : 96);
        TnJob& j = b.job[b.n];
        j.dy = reinterpret_cast<const unsigned short*>(q.dy);
        j.x = reinterpret_cast<const unsigned short*>(q.x);
        j.part = q.partial;
        j.db = q.dbias_partial;
        j.M = (int)q.M;
        j.N = q.N;
        j.K = q.K;
        j.tiles_k = (q.K + tk - 1) / tk;
        j.tiles_n = (q.N + tn - 1) / tn;
        j.rows_per_split = (int)(((q.M + q.splits - 1) / q.splits + MSTEP - 1) / MSTEP * MSTEP);
        j.zero_lo = q.zero_lo;
        j.zero_hi = q.zero_hi;
        j.first_wg = (int)wgs;
        {
            // XCDs per split and the cut of its tile grid (kernel comment): cost of a cut = operand columns fetched by the g XCDs
            const int S = q.splits;
            int g = 1;
            if (S < 8) {
                int p2 = 1;
                while (p2 < S) p2 *= 2;
                g = 8 / p2;
            }
            int best_gn = 1;
            long long best = -1;
            for (int gn = 1; gn <= g; gn *= 2) {
                const int gk = g / gn;
                const long long cost = (long long)((j.tiles_k + gk - 1) / gk) * tk + (long long)((j.tiles_n + gn - 1) / gn) * tn;
                if (best < 0 || cost < best) best = cost, best_gn = gn;
            }
            const int gn = best_gn, gk = g / gn, bk = (j.tiles_k + gk - 1) / gk, bn = (j.tiles_n + gn - 1) / gn;
            const int lanes = 8 / g, rounds = (S + lanes - 1) / lanes;
            j.xmap = g | gn << 8 | bk << 16 | bn << 24;
            j.n_wg = 8 * rounds * bk * bn;
        }
        j.flags = (q.partial_dtype == PSWIN_BF16 ? 1 : 0) | q.splits << 1;
        wgs += j.n_wg;
        if (++b.n == TN_JOBS_MAX || wgs > (1ll << 24)) {
            if (const int rc = flush()) return rc;
        }
    }
    return flush();
}

}  // namespace

extern "C" {

int pswin_gemm_tn_ring_supported(long long M, int N, int K) {
    return M >= 64 && (M + 64) * (long long)(K > N ? K : N) * 2 < 0xFFFFFF00ll && ring_geom(N, K) >= 0;
}

/* row splits for the ring kernel: one workgroup per CU and launch (256 in all, rounded so that no split is shorter than 4 slabs) */
int pswin_gemm_tn_ring_splits(long long M, int N, int K, int target_wgs) {
    if (!pswin_gemm_tn_ring_supported(M, N, K)) return PSWIN_ERR_ARG;
    const int geom = ring_geom(N, K);
    const int tiles = geom == 0 ? (N / 192) * (K / 192) : 1;
    if (target_wgs <= 0) target_wgs = 256;
    int s = target_wgs / tiles;
    const int smax = (int)(M / 256) > 0 ? (int)(M / 256) : 1;
    if (s > smax) s = smax;
    return s < 1 ? 1 : s;
}

int pswin_gemm_tn_ring_jobs(const pswin_tn_job* jobs, int n_jobs, void* stream) {
    PSWIN_CHECK_ARG(jobs && n_jobs > 0);
    for (int i = 0; i < n_jobs; ++i) PSWIN_CHECK_ARG(tn_job_ok(jobs[i]));
    for (int geom = 0; geom < 3; ++geom)
        if (const int rc = launch_tn_jobs(jobs, n_jobs, geom, (hipStream_t)stream)) return rc;
    return PSWIN_OK;
}

int pswin_gemm_tn_ring_bias(const void* dy, const void* x, void* partial, int partial_dtype, float* dbias_partial, int zero_lo, int zero_hi, long long M,
                            int N, int K, int splits, void* stream) {
    pswin_tn_job q;
    q.dy = dy;
    q.x = x;
    q.partial = partial;
    q.dbias_partial = dbias_partial;
    q.M = M;
    q.N = N;
    q.K = K;
    q.splits = splits;
    q.partial_dtype = partial_dtype;
    q.zero_lo = zero_lo;
    q.zero_hi = zero_hi;
    return pswin_gemm_tn_ring_jobs(&q, 1, stream);
}

int pswin_gemm_tn_ring(const void* dy, const void* x, void* partial, int partial_dtype, long long M, int N, int K, int splits, void* stream) {
    return pswin_gemm_tn_ring_bias(dy, x, partial, partial_dtype, nullptr, 0, 0, M, N, K, splits, stream);
}

}  // extern "C"
