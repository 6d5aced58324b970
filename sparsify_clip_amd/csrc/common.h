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
// bf16 epilogues.  GELU(x) = x (0.5 + h(x)) and GELU'(x) = 0.5 + g(x) with h, g odd: both are evaluated as u * r(t) with
// u = clamp(x, -5, 5), t = 2 u^2 / 25 - 1 in [-1, 1] and r a degree-11 polynomial in t (least-squares fit at Chebyshev nodes,
// Horner in the centred variable so the fp32 evaluation does not cancel).  Max abs error against the erf forms, evaluated
// in fp32: 8e-6 for GELU, 1.6e-5 for GELU' (tests/test_gpu_kernels.py) - far below bf16 resolution; beyond |x| = 5 the clamp
// leaves GELU = x * Phi(5) / x * Phi(-5) (|error| < 3e-6 |x|).  No transcendental, and everything runs as packed fp32
// (v_pk_fma_f32, two elements per issue slot): 8.5 VALU slots per element instead of ~23 for the erf form with exp + rcp -
// in the MLP GEMMs' epilogues this VALU work sits on the critical path of every tile.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_odd_series(f32x2 x, const float (&c)[12]) {
    f32x2 u;
    u[0] = __builtin_amdgcn_fmed3f(x[0], -5.0f, 5.0f);
    u[1] = __builtin_amdgcn_fmed3f(x[1], -5.0f, 5.0f);
    const f32x2 t = u * u * 0.08f - 1.0f;
    f32x2 r = t * c[11] + c[10];
#pragma unroll
    for (int k = 9; k >= 0; --k) r = r * t + c[k];
    return u * r + 0.5f;
}
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
    constexpr float c[12] = {1.413636389e-01f, -7.029806329e-02f, 5.153301070e-02f, -4.040101253e-02f, 3.127567917e-02f, -2.353485545e-02f,
                             1.720127132e-02f, -1.047982647e-02f, 4.698281411e-03f, -3.446137215e-03f, 3.396881220e-03f, -1.309042784e-03f};
    return x * gelu_odd_series(x, c);
}
__device__ __forceinline__ f32x2 gelu_grad_fast2(f32x2 x) {
    constexpr float c[12] = {1.421311512e-01f, -7.512982032e-02f, 6.679198904e-02f, -7.137843456e-02f, 7.740823721e-02f, -8.639809652e-02f,
                             9.410022787e-02f, -6.577449719e-02f, 2.253859553e-02f, -3.068536859e-02f, 4.593244559e-02f, -1.953692735e-02f};
    return gelu_odd_series(x, c);
}
__device__ __forceinline__ f32x4 gelu_fast4(f32x4 v) {
    const f32x2 a = gelu_fast2(f32x2{v[0], v[1]}), b = gelu_fast2(f32x2{v[2], v[3]});
    return f32x4{a[0], a[1], b[0], b[1]};
}
// v * GELU'(pre) for four bf16 pre-activations packed in two dwords (little-endian pairs)
__device__ __forceinline__ f32x4 gelu_grad_mul4(f32x4 v, unsigned lo, unsigned hi) {
    const f32x2 a = gelu_grad_fast2(f32x2{__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u)});
    const f32x2 b = gelu_grad_fast2(f32x2{__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)});
    return f32x4{v[0] * a[0], v[1] * a[1], v[2] * b[0], v[3] * b[1]};
}
__device__ __forceinline__ float gelu_fast(float x) { return gelu_fast2(f32x2{x, x})[0]; }
__device__ __forceinline__ float gelu_grad_fast(float x) { return gelu_grad_fast2(f32x2{x, x})[0]; }
#endif
