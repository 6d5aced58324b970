// Shared host/device helpers for libsparsify_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <math.h>
#include "../../include/sparsify_hip.h"

typedef unsigned short bf16_t;  // raw storage
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;

#define SC_WAVE 64

// ----------------------------------------------------------------------------- errors
int sc_set_error(int code, const char* fmt, ...);

#define SC_REQUIRE(cond, code, ...)                         \
    do {                                                    \
        if (!(cond)) return sc_set_error((code), __VA_ARGS__); \
    } while (0)

#define SC_CHECK_LAUNCH()                                                        \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) return sc_set_error((int)e__, "%s:%d launch: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
    } while (0)

#define SC_TRY(expr)                     \
    do {                                 \
        int rc__ = (expr);               \
        if (rc__ != 0) return rc__;      \
    } while (0)

static inline bool sc_aligned(const void* p, size_t a) { return (((uintptr_t)p) & (a - 1)) == 0; }
static inline int64_t sc_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ----------------------------------------------------------------------------- device helpers
#ifdef __HIPCC__
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN) on gfx950
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}

template <typename T> struct io;
template <> struct io<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
    static __device__ __forceinline__ f32x4 ld4(const float* p) { return *(const f32x4*)p; }
    static __device__ __forceinline__ void st4(float* p, f32x4 v) { *(f32x4*)p = v; }
};
template <> struct io<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
    static __device__ __forceinline__ f32x4 ld4(const bf16_t* p) {
        uint2 u = *(const uint2*)p;
        f32x4 r;
        r[0] = __uint_as_float(u.x << 16);
        r[1] = __uint_as_float(u.x & 0xffff0000u);
        r[2] = __uint_as_float(u.y << 16);
        r[3] = __uint_as_float(u.y & 0xffff0000u);
        return r;
    }
    static __device__ __forceinline__ void st4(bf16_t* p, f32x4 v) {
        uint2 u;
        u.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
        u.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
        *(uint2*)p = u;
    }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// exact erf GELU (nn.GELU default) and its derivative
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
// bf16 epilogues: erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below bf16 resolution) - one v_exp + one v_rcp
// instead of erff's long polynomial; exp(-x^2/2) is shared between the cdf and the pdf of the GELU derivative.
__device__ __forceinline__ void gelu_parts_fast(float x, float& cdf, float& pdf_x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float e = __expf(-z * z);                                   // exp(-x^2/2)
    const float t = __frcp_rn(1.0f + 0.3275911f * z);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float tail = 0.5f * poly * e;                               // 0.5 * erfc(|x|/sqrt 2): no cancellation in the tails
    cdf = x >= 0.f ? 1.0f - tail : tail;
    pdf_x = x * 0.39894228040143267794f * e;
}
__device__ __forceinline__ float gelu_fast(float x) {
    float cdf, pdfx;
    gelu_parts_fast(x, cdf, pdfx);
    return x * cdf;
}
__device__ __forceinline__ float gelu_grad_fast(float x) {
    float cdf, pdfx;
    gelu_parts_fast(x, cdf, pdfx);
    return cdf + pdfx;
}
#endif
