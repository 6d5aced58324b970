// Shared host/device helpers for libsparsify_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
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

// A/B knobs of closed experiments (DESIGN.md section 5 lists each with what was measured): the shipped library answers every one of them with
// its default; a -DSC_DEBUG_KNOBS build reads them from the environment again.  The knobs the test-suite exercises (SC_GEMM_NT, SC_GEMM_TN:
// kernel variants that stay in the library for shapes the default kernels do not take) are read with plain getenv.
static inline const char* sc_debug_env(const char* name) {
#ifdef SC_DEBUG_KNOBS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

static inline bool sc_aligned(const void* p, size_t a) { return (((uintptr_t)p) & (a - 1)) == 0; }
static inline int64_t sc_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ----------------------------------------------------------------------------- device helpers
// Deferred final reductions (layernorm.hip; used by block.hip): between sc_reduce_defer_begin and sc_reduce_defer_flush on the calling thread
// the second stages of the column-sum producers (LayerNorm backward, the GEMM / attention bias gradients, sc_colsum) are recorded instead
// of launched, and flush launches them together: one small kernel on the block backward's stream instead of four, each of which the next
// kernel of the stream would wait for.  The partial buffers must stay untouched until the flush; a job's arithmetic is unchanged.
constexpr int SC_REDUCE_JOBS = 6;
struct ScReduceJob {
    const float* partial;
    int nblocks, nwhich, width, accumulate, first_block;
    float *out0, *out1, *out2;
};
struct ScReduceJobs {
    ScReduceJob job[SC_REDUCE_JOBS];
    int n;
};
void sc_reduce_defer_begin(ScReduceJobs* jobs);
int sc_reduce_defer_flush(hipStream_t st);
void sc_reduce_defer_cancel();

#ifdef __HIPCC__
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN) on gfx950
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}

// two floats -> one dword of two bf16 (low half = a): ONE v_cvt_pk_bf16_f32.  Written as (unsigned)f32_to_bf16(a) | (unsigned)f32_to_bf16(b) << 16
// the compiler emits two conversions, a shift and an or - four vector instructions per pair in epilogues that are bound by the vector ALU.
typedef __attribute__((ext_vector_type(2))) float sc_f32v2;
typedef __attribute__((ext_vector_type(2))) __bf16 sc_bf16v2;
__device__ __forceinline__ unsigned pack2_bf16(float a, float b) {
    const sc_f32v2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, sc_bf16v2));
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
        u.x = pack2_bf16(v[0], v[1]);
        u.y = pack2_bf16(v[2], v[3]);
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
// u = clamp(x, -U, U), t = 2 u^2 / U^2 - 1 in [-1, 1] and r a polynomial in t (weighted minimax fit of u (r - exact) at Chebyshev
// nodes, tools/fit_gelu_series.py; Horner in the centred variable so the fp32 evaluation does not cancel).  The outputs are rounded
// to bf16 (half an ulp = 2^-8 relative = 3.9e-3), so the series only need a fraction of that:
//   GELU : degree 6, U = 3.8: |Phi error| <= 6.4e-5 everywhere, i.e. relative error <= 1.3e-4 (1/30 of the rounding) for x > 0 and
//          |error| <= 6.4e-5 |x| for x < 0 (beyond the clamp GELU = x Phi(+-3.8), |error| < 7.3e-5 |x|)
//   GELU': degree 7, U = 4.0: |error| <= 2.6e-4 (1/15 of the rounding of a value near 1)
// (tests/test_gpu_kernels.py::test_gemm_bf16_epilogue_gelu_series_accuracy; rounds 1-2 used degree 11 on [-5, 5], 8e-6 / 1.6e-5,
// 16 VALU operations per element against 11 now).  No transcendental: on gfx950 v_exp_f32 / v_rcp_f32 hold the vector pipe for four
// plain operations each, so the erf / exp forms cost more.  In the MLP GEMMs' epilogues this work is bound by the VALU pipe itself
// (two waves per SIMD, 32 lanes per clock) and sits on the critical path of every tile.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int N>
__device__ __forceinline__ f32x2 gelu_odd_series(f32x2 x, const float (&c)[N], float clamp, float scale) {
    f32x2 u;
    u[0] = __builtin_amdgcn_fmed3f(x[0], -clamp, clamp);
    u[1] = __builtin_amdgcn_fmed3f(x[1], -clamp, clamp);
    const f32x2 t = u * u * scale - 1.0f;
    f32x2 r = t * c[N - 1] + c[N - 2];
#pragma unroll
    for (int k = N - 3; k >= 0; --k) r = r * t + c[k];
    return u * r + 0.5f;
}
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
    constexpr float c[7] = {1.847486975e-01f, -8.713941217e-02f, 5.531427544e-02f, -3.298223234e-02f, 1.987116002e-02f, -1.342840381e-02f,
                            5.192545850e-03f};
    return x * gelu_odd_series(x, c, 3.8f, 2.0f / (3.8f * 3.8f));
}
__device__ __forceinline__ f32x2 gelu_grad_fast2(f32x2 x) {
    constexpr float c[8] = {1.831712853e-01f, -1.137763957e-01f, 1.175296832e-01f, -1.133965505e-01f, 8.304241140e-02f, -7.422325352e-02f,
                            7.714811401e-02f, -3.443386059e-02f};
    return gelu_odd_series(x, c, 4.0f, 0.125f);
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
