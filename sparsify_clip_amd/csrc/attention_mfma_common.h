// Shared device helpers of the bf16 MFMA attention kernels (attention_mfma.hip, attention_mfma_long.hip).
#pragma once
#include "common.h"

namespace attn {

constexpr int HD = 64;
constexpr int LDR = 72;   // LDS row stride in bf16 elements (144 B)

typedef __attribute__((address_space(3))) bf16x4* ltr_t;

static __device__ __forceinline__ bf16x4 tr_read(const bf16_t* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)p); }

// fragment of the transposed image for k-step s: element e <-> row 32s + 4g + e (e<4) / 32s + 16 + 4g + (e-4)
template <bool HI_VALID>
static __device__ __forceinline__ bf16x8 tr_frag(const bf16_t* img, int s, int col16, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const bf16_t* a = img + (32 * s + 4 * g + q) * LDR + col16 + 4 * p;
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
    return *(const bf16x8*)(img + (tile * 16 + (lane & 15)) * LDR + 32 * ks + 8 * (lane >> 4));
}
static __device__ __forceinline__ bf16x8 row_frag_global(const bf16_t* base, int64_t ld, int tile, int ks, int lane, int S) {
    const int r = min(tile * 16 + (lane & 15), S - 1);
    return *(const bf16x8*)(base + (int64_t)r * ld + 32 * ks + 8 * (lane >> 4));
}

static __device__ __forceinline__ bf16x8 pack_frag(const f32x4& lo, const f32x4& hi) {
    bf16x8 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[e] = (short)f32_to_bf16(lo[e]); f[4 + e] = (short)f32_to_bf16(hi[e]); }
    return f;
}

// copy a [S][64] bf16 head slice (row stride ld) into an LDS image of NT*16 rows; rows >= S are zero
template <int NT>
static __device__ __forceinline__ void stage_head(bf16_t* img, const bf16_t* src, int64_t ld, int S, int lane) {
#pragma unroll
    for (int it = 0; it < NT * 2; ++it) {
        const int r = it * 8 + (lane >> 3), c = (lane & 7) * 8;
        uint4 v = {0u, 0u, 0u, 0u};
        if (r < S) v = *(const uint4*)(src + (int64_t)r * ld + c);
        *(uint4*)(img + r * LDR + c) = v;
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
        *(uint4*)(img + r * LDR + c) = v;
    }
}

static __device__ __forceinline__ float group_max(float v) {   // across the 4 lane groups that share one query row
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
static __device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)


// stage with all `nthreads` threads of the workgroup (block-per-head kernels)
static __device__ __forceinline__ void stage_head_block(bf16_t* img, const bf16_t* src, int64_t ld, int S, int rows_pad, int tid, int nthreads) {
    for (int i = tid; i < rows_pad * 8; i += nthreads) {
        const int r = i >> 3, c = (i & 7) * 8;
        uint4 v = {0u, 0u, 0u, 0u};
        if (r < S) v = *(const uint4*)(src + (int64_t)r * ld + c);
        *(uint4*)(img + r * LDR + c) = v;
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
            float x = acc[dt][r];
            x += __shfl_xor(x, 1, 64); x += __shfl_xor(x, 2, 64); x += __shfl_xor(x, 4, 64); x += __shfl_xor(x, 8, 64);
            acc[dt][r] = x;
        }
}
