// Shared device helpers of the bf16 MFMA attention kernels (attention_mfma.hip, attention_mfma_long.hip).
#pragma once
#include "common.h"

namespace attn {

constexpr int HD = 64;
// LDS images: [rows][64] bf16 in plain 128-byte rows, the eight 16-byte chunks of a row XOR-swizzled by the row (chunk ^ (row & 7)).
// By the bank rules of MI355X_MICROARCH.md (LDS): the ds_read_b128 row read of an MFMA operand (16 rows x one chunk per 16-lane group,
// groups mixed as {0-3, 12-15, 20-27} ...) and the ds_read_b64_tr_b16 transposed read (8 rows x 4 half-chunks per 32-lane half) both touch
// 64 distinct banks.  Rounds 1-2 padded the rows to 144 bytes: 2-way conflicts on both kinds of read, 38-40 % of the LDS-active cycles
// (profiles/r02_attention_pmc_counters.txt, r03_attention_pmc_counters_before_swizzle.txt), and 12.5 % more LDS per image.
constexpr int LDR = 64;   // LDS row stride in bf16 elements (128 B)
static __device__ __forceinline__ int img_off(int row, int chunk) { return row * LDR + ((chunk ^ (row & 7)) << 3); }   // element offset of a 16-byte chunk

typedef __attribute__((address_space(3))) bf16x4* ltr_t;

static __device__ __forceinline__ bf16x4 tr_read(const bf16_t* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)p); }

// fragment of the transposed image for k-step s: element e <-> row 32s + 4g + e (e<4) / 32s + 16 + 4g + (e-4)
template <bool HI_VALID>
static __device__ __forceinline__ bf16x8 tr_frag(const bf16_t* img, int s, int col16, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const bf16_t* a = img + img_off(32 * s + 4 * g + q, (col16 >> 3) + (p >> 1)) + 4 * (p & 1);   // rows 16 apart share the swizzle: + 16 * LDR below
    const bf16x4 lo = tr_read(a);
    bf16x4 hi = {0, 0, 0, 0};
    if (HI_VALID) hi = tr_read(a + 16 * LDR);
    bf16x8 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[e] = lo[e]; f[4 + e] = hi[e]; }
    return f;
}

// row fragment (A or B operand with the row on the lane): rows tile*16 + (lane&15), k = 32*ks + 8*(lane>>4) + e
static __device__ __forceinline__ bf16x8 row_frag_lds(const bf16_t* img, int tile, int ks, int lane) {
    return *(const bf16x8*)(img + img_off(tile * 16 + (lane & 15), 4 * ks + (lane >> 4)));
}
static __device__ __forceinline__ bf16x8 row_frag_global(const bf16_t* base, int64_t ld, int tile, int ks, int lane, int S) {
    const int r = min(tile * 16 + (lane & 15), S - 1);
    return *(const bf16x8*)(base + (int64_t)r * ld + 32 * ks + 8 * (lane >> 4));
}

static __device__ __forceinline__ bf16x8 pack_frag(const f32x4& lo, const f32x4& hi) {
    const uint4 u = {pack2_bf16(lo[0], lo[1]), pack2_bf16(lo[2], lo[3]), pack2_bf16(hi[0], hi[1]), pack2_bf16(hi[2], hi[3])};   // four v_cvt_pk_bf16_f32
    return __builtin_bit_cast(bf16x8, u);
}

// copy a [S][64] bf16 head slice (row stride ld) into an LDS image of NT*16 rows; rows >= S are zero
template <int NT>
static __device__ __forceinline__ void stage_head(bf16_t* img, const bf16_t* src, int64_t ld, int S, int lane) {
#pragma unroll
    for (int it = 0; it < NT * 2; ++it) {
        const int r = it * 8 + (lane >> 3), c = (lane & 7) * 8;
        uint4 v = {0u, 0u, 0u, 0u};
        if (r < S) v = *(const uint4*)(src + (int64_t)r * ld + c);
        *(uint4*)(img + img_off(r, c >> 3)) = v;
    }
}

// the same by the WPH waves of a head: wave `part` copies every WPH-th group of 8 rows
template <int NT, int WPH>
static __device__ __forceinline__ void stage_head_part(bf16_t* img, const bf16_t* src, int64_t ld, int S, int lane, int part) {
#pragma unroll
    for (int it2 = 0; it2 < (2 * NT + WPH - 1) / WPH; ++it2) {
        const int it = WPH * it2 + part;
        if (it >= 2 * NT) break;
        const int r = it * 8 + (lane >> 3), c = (lane & 7) * 8;
        uint4 v = {0u, 0u, 0u, 0u};
        if (r < S) v = *(const uint4*)(src + (int64_t)r * ld + c);
        *(uint4*)(img + img_off(r, c >> 3)) = v;
    }
}

// Reductions across the 4 lane groups (lane >> 4) that share one query row, on the vector ALU: v_permlane16_swap exchanges the odd
// 16-lane rows of its first operand with the even rows of its second, v_permlane32_swap the upper half of the first with the lower
// half of the second - with both operands the same value every lane ends up holding its own and its partner's value.  (The shuffles
// of rounds 1-2 were ds_bpermute: an LDS round trip in the middle of every softmax dependency chain.)
static __device__ __forceinline__ void group_pair16(float v, float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
static __device__ __forceinline__ void group_pair32(float v, float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
static __device__ __forceinline__ float group_max(float v) {
    float a, b;
    group_pair16(v, a, b); v = fmaxf(a, b);
    group_pair32(v, a, b); return fmaxf(a, b);
}
static __device__ __forceinline__ float group_sum(float v) {   // same value in all four lanes: a + b is commutative, the pairing is fixed
    float a, b;
    group_pair16(v, a, b); v = a + b;
    group_pair32(v, a, b); return a + b;
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)


// stage with all `nthreads` threads of the workgroup (block-per-head kernels)
static __device__ __forceinline__ void stage_head_block(bf16_t* img, const bf16_t* src, int64_t ld, int S, int rows_pad, int tid, int nthreads) {
    for (int i = tid; i < rows_pad * 8; i += nthreads) {
        const int r = i >> 3, c = (i & 7) * 8;
        uint4 v = {0u, 0u, 0u, 0u};
        if (r < S) v = *(const uint4*)(src + (int64_t)r * ld + c);
        *(uint4*)(img + img_off(r, c >> 3)) = v;
    }
}

}  // namespace attn

// Fused in_proj bias gradient: running column sums of the dq / dk / dv tiles a wave produces (values as stored, i.e. bf16-rounded),
// reduced over the 16 row lanes at the end.  acc[dt][r] belongs to column 16 dt + 4 (lane >> 4) + r of the head.
__device__ __forceinline__ void cs_add(f32x4 (&acc)[4], const f32x4 (&v)[4], bool live) {
    if (!live) return;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[dt][r] += bf16_to_f32(f32_to_bf16(v[dt][r]));
}
__device__ __forceinline__ void cs_rows(f32x4 (&acc)[4]) {   // sum over lane bits 0..3 (the 16 rows of a tile), fixed order
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = acc[dt][r];   // DPP butterflies inside the 16-lane row (no LDS round trip): xor 1, xor 2, mirror of 8, mirror of 16
            x += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(x), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
            x += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(x), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
            x += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(x), 0x141, 0xF, 0xF, true));   // row_half_mirror
            x += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(x), 0x140, 0xF, 0xF, true));   // row_mirror
            acc[dt][r] = x;
        }
}
