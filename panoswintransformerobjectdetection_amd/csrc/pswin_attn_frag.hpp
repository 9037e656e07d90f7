// Device helpers shared by the window-attention kernels (pswin_attn.hip: attention core; pswin_fused.hip: the per-window
// qkv -> attention -> proj kernel): MFMA operand fragments, LDS images with transposed reads, cross-lane reductions, and the
// PanoSwin score bias (HOT:241-272).  HOT = mmdet/models/backbones/simple_panoswin_transformer.py of the reference.
#pragma once
#include "pswin_common.hpp"

namespace {

using namespace pswin;

constexpr int TOK = PSWIN_WTOK;    // 49
constexpr int PADT = PSWIN_WPAD;   // 64
constexpr int HD = PSWIN_HEAD_DIM; // 32
constexpr float LOG2E = 1.4426950408889634f;

// ---------------------------------------------------------------------------------------------
// fragments: 8 head-dim (or key) elements of one row, the unit both MFMA flavours contract over
// ---------------------------------------------------------------------------------------------
template <int DT>
struct Frag;
template <>
struct Frag<PSWIN_BF16> {
    bf16x8 v;
};
template <>
struct Frag<PSWIN_F32> {
    float v[8];
};

template <int DT>
__device__ inline Frag<DT> zero_frag() {
    Frag<DT> f;
    if constexpr (DT == PSWIN_BF16) {
        u32x4 z = {0u, 0u, 0u, 0u};
        f.v = __builtin_bit_cast(bf16x8, z);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) f.v[i] = 0.f;
    }
    return f;
}

// 8 consecutive elements starting at element offset `off` (16-byte aligned)
template <int DT>
__device__ inline Frag<DT> load_frag(const void* base, size_t off, bool valid) {
    Frag<DT> f = zero_frag<DT>();
    if (valid) {
        if constexpr (DT == PSWIN_BF16) {
            u32x4 raw = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(base) + off);
            f.v = __builtin_bit_cast(bf16x8, raw);
        } else {
            const float* p = reinterpret_cast<const float*>(base) + off;
            f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f.v[i] = a[i];
                f.v[4 + i] = b[i];
            }
        }
    }
    return f;
}

// Operand addressing: buffer instructions.  Per image a wave builds one buffer resource per tensor on the SALU (base
// = row 0 of the window, column 0 of the head; range = rows 0..48 of that head), and every lane keeps ONE loop-invariant
// 32-bit byte offset per access slot: `buffer_load_dwordx4 v, v_off, s[rsrc], 0 offen`.  Against flat 64-bit VGPR
// addresses this (a) frees ~2 VGPRs per slot and the VALU adds that rebuilt them per image (the backward kernel was
// spilling its pointers; the reloads serialised on vmcnt(0) behind the prefetch of the next image), and (b) makes the
// padded rows 49..63 free: their offsets fall outside the resource's range, so loads return zeros and stores are
// dropped by the hardware range check -- no branches, no exec masking.
template <int DT>
constexpr int ES = (DT == PSWIN_BF16) ? 2 : 4;
using rsrc_t = __amdgpu_buffer_rsrc_t;

// resource for the [49 rows][HD] head slice starting at base + byte_off, rows ld elements apart
template <int DT>
__device__ inline rsrc_t window_rsrc(const void* base, size_t byte_off, int ld) {
    char* p = const_cast<char*>(reinterpret_cast<const char*>(base)) + byte_off;
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, ((TOK - 1) * ld + HD) * ES<DT>, 0x00020000);
}

// byte offset of the 8-element group g of row `row` within the window's head slice
template <int DT>
__device__ inline unsigned row_off(int row, int ld, int g) {
    return (unsigned)(row * ld + 8 * g) * ES<DT>;
}

template <int DT>
__device__ inline Frag<DT> load_frag_at(rsrc_t rs, unsigned boff) {
    Frag<DT> f;
    if constexpr (DT == PSWIN_BF16) {
        u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rs, boff, 0, 0);
        f.v = __builtin_bit_cast(bf16x8, raw);
    } else {
        const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, boff, 0, 0));
        const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, boff + 16u, 0, 0));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f.v[i] = a[i];
            f.v[4 + i] = b[i];
        }
    }
    return f;
}

// two accumulator quads -> the 8-element operand of the next MFMA (k order: lo[0..3], hi[0..3])
template <int DT>
__device__ inline Frag<DT> pack_frag(f32x4 lo, f32x4 hi) {
    Frag<DT> f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if constexpr (DT == PSWIN_BF16) {
            f.v[i] = (__bf16)lo[i];
            f.v[4 + i] = (__bf16)hi[i];
        } else {
            f.v[i] = lo[i];
            f.v[4 + i] = hi[i];
        }
    }
    return f;
}

// acc[16x16] += A[16 x 32] . B[32 x 16]; lane (c = lane & 15, g = lane >> 4) supplies row/column c of A/B and the
// 8 contraction elements of its group g.  bf16: one v_mfma_f32_16x16x32_bf16.  f32: eight v_mfma_f32_16x16x4_f32,
// step s contracting element s of every group (same pairing on both operands, so any k order is valid).
template <int DT>
__device__ inline f32x4 mma32(const Frag<DT>& a, const Frag<DT>& b, f32x4 acc) {
    if constexpr (DT == PSWIN_BF16) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[s], b.v[s], acc, 0, 0, 0);
        return acc;
    }
}

// ---------------------------------------------------------------------------------------------
// LDS images: row-major [64 rows][COLS] tiles written row-wise (8 or 4 elements per lane) and read
// TRANSPOSED: lane (c, g) receives, for column col0 + c, the 8 rows {R0 + 0..3, R0 + 16 + 0..3}.
// ---------------------------------------------------------------------------------------------
template <int DT, int COLS>
struct LdsImg {
    // bf16 [.][32] images: dense 64-byte rows whose four 16-byte chunks are XOR-swizzled by the row (img32_off below);
    // other bf16 images: 16 bytes of row padding; f32 rows: 4 floats of padding so that the 4 lane groups of a column
    // read hit different banks.
    static constexpr int LD = (DT == PSWIN_BF16) ? (COLS == 32 ? 32 : COLS + 8) : COLS + 4;
    static constexpr int BYTES = PADT * LD * (DT == PSWIN_BF16 ? 2 : 4);
};

// Byte offset of 16-byte chunk `chunk16` (0..3) of row `row` in a bf16 [.][32] image.  Dense power-of-two rows are the
// worst case for the LDS banks (SQ_LDS_BANK_CONFLICT measured 5x the conflict-free LDS time in the backward kernel):
// a ds_write_b128 of 8 rows hits 2 bank groups (4-way), a ds_read_b64_tr_b16 half-wave (8 rows x 32 B) 2-way.  XOR-ing
// the chunk with bits 1..2 of the row makes both conflict-free: rows of equal parity get 4 different chunks, and rows
// r, r + 4 (same quarter of the 64 banks) use different 32-byte halves.  Unchanged by row + 16.
__device__ inline int img32_off(int row, int chunk16) { return row * 64 + ((chunk16 ^ ((row >> 1) & 3)) << 4); }

template <int DT, int COLS>
__device__ inline void lds_write_frag(char* img, int row, int col, const Frag<DT>& f) {
    constexpr int LD = LdsImg<DT, COLS>::LD;
    if constexpr (DT == PSWIN_BF16 && COLS == 32) {
        *reinterpret_cast<bf16x8*>(img + img32_off(row, col >> 3)) = f.v;
    } else if constexpr (DT == PSWIN_BF16) {
        *reinterpret_cast<bf16x8*>(img + ((size_t)row * LD + col) * 2) = f.v;
    } else {
        float* p = reinterpret_cast<float*>(img) + (size_t)row * LD + col;
        f32x4 a = {f.v[0], f.v[1], f.v[2], f.v[3]}, b = {f.v[4], f.v[5], f.v[6], f.v[7]};
        *reinterpret_cast<f32x4*>(p) = a;
        *reinterpret_cast<f32x4*>(p + 4) = b;
    }
}

// 4 consecutive elements of one row (an accumulator quad) -> LDS
template <int DT, int COLS>
__device__ inline void lds_write_quad(char* img, int row, int col, f32x4 q) {
    constexpr int LD = LdsImg<DT, COLS>::LD;
    if constexpr (DT == PSWIN_BF16) {
        bf16x4 b = {(__bf16)q[0], (__bf16)q[1], (__bf16)q[2], (__bf16)q[3]};
        *reinterpret_cast<bf16x4*>(img + ((size_t)row * LD + col) * 2) = b;
    } else {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(img) + (size_t)row * LD + col) = q;
    }
}

// transposed operand read.  EXEC must be all ones (ds_read_b64_tr_b16 gathers across the 16-lane group).
template <int DT, int COLS>
__device__ inline Frag<DT> lds_read_tr(const char* img, int R0, int col0, int c, int g) {
    constexpr int LD = LdsImg<DT, COLS>::LD;
    Frag<DT> f;
    if constexpr (DT == PSWIN_BF16) {
        // lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of the 4 x 16 block; it receives
        // column (4q+p) of the 4 rows.
        const int q = c >> 2, p = c & 3;
        typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
        const char* a0;
        if constexpr (COLS == 32) {
            const int k8 = (col0 >> 2) + p;               // 8-byte unit within the row
            a0 = img + img32_off(R0 + q, k8 >> 1) + ((k8 & 1) << 3);
        } else {
            a0 = img + ((size_t)(R0 + q) * LD + col0 + 4 * p) * 2;
        }
        const char* a1 = a0 + (size_t)16 * LD * 2;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a1));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        f.v = __builtin_bit_cast(bf16x8, both);
    } else {
        const float* base = reinterpret_cast<const float*>(img) + col0 + c;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) f.v[jj] = base[(size_t)(R0 + 16 * (jj >> 2) + (jj & 3)) * LD];
    }
    (void)g;
    return f;
}

// 4 consecutive output elements of one row
template <int DT>
__device__ inline void store_quad(void* base, size_t off, f32x4 q) {
    store4<DT>(base, off, q);
}

__device__ inline void swap16_u32(unsigned& a, unsigned& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }

// A lane (c, g) holds, for one output row, the head-dim quads q0 = d[4g .. 4g+3] and q1 = d[16+4g .. 16+4g+3] (the two
// 16-wide MFMA tiles).  Exchanging q1 of the even groups with q0 of the odd groups (lanes 16 apart: one
// v_permlane16_swap per dword) leaves every lane with 8 CONTIGUOUS elements, d0 = 8 (g >> 1) + 16 (g & 1), so the row
// is written with half as many, twice as wide stores (8-byte bf16 stores are store-issue bound: MI355X guide T21).
// Every lane takes part in the exchange; the stores of padded rows (>= 49) are dropped by the buffer range check.
// row_boff: byte offset of the lane's row within the window's head slice (see window_rsrc).
template <int DT>
__device__ inline void store_row8_at(rsrc_t rs, unsigned row_boff, int g, f32x4 q0, f32x4 q1) {
    const int d0 = 8 * (g >> 1) + 16 * (g & 1);
    if constexpr (DT == PSWIN_BF16) {
        unsigned a0 = pack2_bf16(q0[0], q0[1]), a1 = pack2_bf16(q0[2], q0[3]), b0 = pack2_bf16(q1[0], q1[1]), b1 = pack2_bf16(q1[2], q1[3]);
        swap16_u32(a0, b0);
        swap16_u32(a1, b1);
        const u32x4 v = {a0, a1, b0, b1};
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, row_boff + (unsigned)d0 * 2u, 0, 0);
    } else {
        // f32 quads are already 16-byte stores: no exchange needed
        (void)d0;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, q0), rs, row_boff + 16u * g, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, q1), rs, row_boff + 64u + 16u * g, 0, 0);
    }
}

// Cross-lane reductions on the VALU (no ds_bpermute round trips through the LDS crossbar).
// v_permlane16_swap(a, b) exchanges the odd 16-lane rows of a with the even rows of b; with a = b = v the two results
// are [r0 r0 r2 r2] and [r1 r1 r3 r3], so combining them reduces over lane ^ 16.  v_permlane32_swap likewise for ^ 32.
// Written as inline asm: hipcc (ROCm 7.2) deletes the combine after __builtin_amdgcn_permlaneNN_swap(x, x) (it
// treats the two results as equal; seen in the ISA, outputs wrong by the missing reduction).  The s_nop covers the
// VALU-write -> permlane-read hazard the compiler would otherwise pad itself.
__device__ inline void swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ inline void swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ inline float group_max(float v) {   // across the 4 lane groups (lanes c, c+16, c+32, c+48)
    float a = v, b = v;
    swap16(a, b);
    a = fmaxf(a, b);
    b = a;
    swap32(a, b);
    return fmaxf(a, b);
}
__device__ inline float group_sum(float v) {
    float a = v, b = v;
    swap16(a, b);
    a = a + b;
    b = a;
    swap32(a, b);
    return a + b;
}
template <int CTRL>
__device__ inline float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ inline float row16_sum(float v) {   // across the 16 lanes of a group: quad xor 1, xor 2, then mirrors
    v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);   // row_half_mirror: the other quad of the 8 (all 4 lanes of a quad agree by now)
    return dpp_add<0x140>(v);   // row_mirror: the other half of the 16
}

constexpr int NBINS = (2 * PSWIN_WS - 1) * (2 * PSWIN_WS - 1);   // 169
constexpr int TABP = 176;                                         // padded table length

__device__ inline int rel_a(int i) { return 13 * (i / PSWIN_WS) + i % PSWIN_WS + 84; }   // idx(i, j) = rel_a(i) - rel_b(j)
__device__ inline int rel_b(int j) { return 13 * (j / PSWIN_WS) + j % PSWIN_WS; }

// bias quad for query index base qi (4 consecutive when QUERY_ON_REGS) / key index base kj.
// QUERY_ON_REGS = false (forward):  i = qi fixed, j = kj + e          tile row = i, quad along j
// QUERY_ON_REGS = true  (backward): i = qi + e,   j = kj fixed        tile row = j (transposed tiles), quad along i
// The result is bias / scale: the kernels run the MFMAs on the UNSCALED q (score' = q.k + bias/scale) and fold the
// scale into the exp2 argument, which removes the per-image q*scale pass.  Branch-free (selects only).
// In two halves, so that a kernel can request the distance / mask quads of ALL its tiles (and its table columns) before it
// waits for the first: built one quad at a time, each quad's loads were a memory round trip of their own in front of the
// batch loop (8-16 of them per work item, as long as the loop itself at 4-8 images per item).
struct BiasRaw {
    f32x4 d, m;
};
// Buffer loads: a null tile is a resource of zero records (the loads return 0: no branch around them) and the per-lane address
// is one 32-bit offset -- with 64-bit pointers the register allocator reused a pending load's destination for the next address
// and the kernel waited for memory in the middle of its request burst.
__device__ inline __amdgpu_buffer_rsrc_t tile_rsrc(const float* tile) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tile), 0, tile ? PADT * PADT * 4 : 0, 0x00020000);
}
template <bool QUERY_ON_REGS>
__device__ inline BiasRaw bias_fetch(__amdgpu_buffer_rsrc_t dres, __amdgpu_buffer_rsrc_t mres, int qi, int kj) {
    const int row = QUERY_ON_REGS ? kj : qi, col = QUERY_ON_REGS ? qi : kj;
    BiasRaw r;
    r.d = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dres, (row * PADT + col) * 4, 0, 0));
    r.m = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(mres, (row * PADT + col) * 4, 0, 0));
    return r;
}
template <bool QUERY_ON_REGS>
__device__ inline BiasRaw bias_fetch(const float* dtile, const float* mtile, int qi, int kj) {
    return bias_fetch<QUERY_ON_REGS>(tile_rsrc(dtile), tile_rsrc(mtile), qi, kj);
}
// the table entries of a quad: 4 (alpha, beta) pairs out of the LDS copy of the head's table columns.  Read for ALL quads of a
// kernel before the first is combined (bias_lookup, then bias_from): looked up inside the arithmetic, the 32-64 lookups of a
// work item were as many LDS round trips in a row
struct BiasTab {
    f32x4 a, b;
};
template <bool QUERY_ON_REGS>
__device__ inline BiasTab bias_lookup(const float* tab_a, const float* tab_b, int qi, int kj) {
    BiasTab t;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = QUERY_ON_REGS ? qi + e : qi, j = QUERY_ON_REGS ? kj : kj + e;
        const bool real = (i < TOK) & (j < TOK);
        int idx = rel_a(i) - rel_b(j);
        idx = real ? idx : 0;
        t.a[e] = tab_a[idx];
        t.b[e] = tab_b[idx];
    }
    return t;
}
template <bool QUERY_ON_REGS>
__device__ inline f32x4 bias_from(const BiasRaw& raw, const BiasTab& tab, int qi, int kj, float inv_scale) {
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = QUERY_ON_REGS ? qi + e : qi, j = QUERY_ON_REGS ? kj : kj + e;
        // same rounding sequence as the reference: (d * alpha + beta) [+ mask]   (HOT:255-256, 294, 301).  No branches: without
        // a distance tile d = 0 and the staged alpha column is 0 (0 * 0 + beta = beta), without a mask tile m = 0.
        float val = __fadd_rn(__fmul_rn(raw.d[e], tab.a[e]), tab.b[e]);
        val = __fadd_rn(val, raw.m[e]);
        val *= inv_scale;
        val = (i < TOK) ? val : 0.f;                 // padded query row: discarded
        r[e] = (j < TOK) ? val : -INFINITY;          // padded key: never receives weight
    }
    return r;
}
template <bool QUERY_ON_REGS>
__device__ inline f32x4 bias_quad(const float* dtile, const float* mtile, const float* tab_a, const float* tab_b,
                                  int qi, int kj, float inv_scale) {
    return bias_from<QUERY_ON_REGS>(bias_fetch<QUERY_ON_REGS>(dtile, mtile, qi, kj), bias_lookup<QUERY_ON_REGS>(tab_a, tab_b, qi, kj), qi, kj,
                                    inv_scale);
}
}  // namespace
