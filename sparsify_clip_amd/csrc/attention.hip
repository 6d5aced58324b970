// Small-sequence multi-head attention (S = 50 / 77, head dim 64) on packed qkv [B*S, 3W]
// (nn.MultiheadAttention in_proj layout: q | k | v, head h = columns h*64 .. h*64+63 of each third).
// One workgroup per (batch, head): the whole head (Q, K, V, and for the backward dO, P, dS) lives in LDS as fp32,
// softmax statistics are wave reductions, nothing of size S x S ever reaches HBM.  fp32 arithmetic for both
// storage dtypes, so the same kernel serves the fp32 parity path and the bf16 path.
// The text tower uses the additive causal mask of open_clip (-inf above the diagonal).
#include "common.h"
#include "gemm_epilogue.h"
#include <stdlib.h>

int sc_attention_mfma_fwd(const void* qkv, void* out, int64_t batch, int64_t seq, int64_t width, int64_t heads, int causal, hipStream_t st);
int sc_attention_mfma_bwd(const void* qkv, const void* d_out, void* d_qkv, int64_t batch, int64_t seq, int64_t width, int64_t heads, int causal,
                          float* cs_part, hipStream_t st);
int sc_attention_long_fwd(const void* qkv, void* out, int64_t batch, int64_t seq, int64_t width, int64_t heads, int causal, hipStream_t st, float* lse);
int sc_attention_long_bwd(const void* qkv, const void* d_out, void* d_qkv, int64_t batch, int64_t seq, int64_t width, int64_t heads, int causal,
                          float* cs_part, hipStream_t st, const void* fwd_out, const float* lse);
bool sc_attention_long_uses_stats(int64_t seq);

int sc_colsum_reduce(const float* partial, int nslab, int64_t n, float* out, int accumulate, hipStream_t st);
extern "C" int sc_colsum(const void* x, int dtype, int64_t rows, int64_t n, int64_t ld, float* out, int accumulate, void* ws, size_t ws_bytes,
                         void* stream);

namespace {

// SC_ATTENTION=valu forces the fp32-VALU kernels for bf16 too (A/B runs)
bool use_mfma() {
    static const bool on = [] { const char* e = sc_debug_env("SC_ATTENTION"); return !(e && e[0] == 'v'); }();
    return on;
}
// SC_ATTENTION_SHORT=2: the recompute (workgroup-per-head) kernels for short sequences too (A/B runs)
bool short_recompute() {
    static const bool on = [] { const char* e = sc_debug_env("SC_ATTENTION_SHORT"); return e && e[0] == '2'; }();
    return on;
}

constexpr int HD = 64;        // head dim
constexpr int HDP = HD + 4;   // padded LDS row (floats): 272-B stride -> conflict-free ds_read_b128 across rows

template <typename T>
__device__ __forceinline__ void load_head(float* dst, const T* src, int64_t ld, int S, int tid) {
    // dst[S][HDP] <- src[s*ld + 0..63]
    for (int i = tid; i < S * (HD / 4); i += 256) {
        const int s = i >> 4, c = (i & 15) * 4;
        *(f32x4*)(dst + s * HDP + c) = io<T>::ld4(src + (int64_t)s * ld + c);
    }
}

__device__ __forceinline__ float dot64(const float* a, const float* b) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
        const f32x4 x = *(const f32x4*)(a + c), y = *(const f32x4*)(b + c);
        s += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
    }
    return s;
}

// softmax row i of (scale * q_i . k_j) into p0 (j = lane) and p1 (j = 64 + lane); masked entries are 0
template <bool CAUSAL>
__device__ __forceinline__ void softmax_row(const float* Qs, const float* Ks, int S, int i, int lane, float scale, float& p0, float& p1) {
    const int j0 = lane, j1 = 64 + lane;
    const bool v0 = j0 < S && (!CAUSAL || j0 <= i), v1 = j1 < S && (!CAUSAL || j1 <= i);
    const float s0 = v0 ? dot64(Qs + i * HDP, Ks + j0 * HDP) * scale : -INFINITY;
    const float s1 = v1 ? dot64(Qs + i * HDP, Ks + j1 * HDP) * scale : -INFINITY;
    const float m = wave_max(fmaxf(s0, s1));
    p0 = v0 ? expf(s0 - m) : 0.f;
    p1 = v1 ? expf(s1 - m) : 0.f;
    const float inv = 1.0f / wave_sum(p0 + p1);
    p0 *= inv;
    p1 *= inv;
}

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(256) void attention_fwd_kernel(const T* qkv, T* out, int S, int W, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Qs = lds;
    float* Ks = Qs + S * HDP;
    float* Vs = Ks + S * HDP;
    float* Ps = Vs + S * HDP;   // [4][128]
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t ld = 3 * (int64_t)W;
    const T* base = qkv + (int64_t)b * S * ld + h * HD;
    load_head<T>(Qs, base, ld, S, tid);
    load_head<T>(Ks, base + W, ld, S, tid);
    load_head<T>(Vs, base + 2 * W, ld, S, tid);
    __syncthreads();
    const int iters = (S + 3) / 4;
    for (int it = 0; it < iters; ++it) {
        const int i = it * 4 + w;
        if (i < S) {
            float p0, p1;
            softmax_row<CAUSAL>(Qs, Ks, S, i, lane, scale, p0, p1);
            Ps[w * 128 + lane] = p0;
            Ps[w * 128 + 64 + lane] = p1;
        }
        __syncthreads();
        if (i < S) {
            const int jn = CAUSAL ? i + 1 : S;
            float o = 0.f;
            for (int j = 0; j < jn; ++j) o += Ps[w * 128 + j] * Vs[j * HDP + lane];
            io<T>::st(out + ((int64_t)b * S + i) * W + h * HD + lane, o);
        }
    }
}

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(256) void attention_bwd_kernel(const T* qkv, const T* d_out, T* d_qkv, int S, int W, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int SP = S + 1;
    float* Qs = lds;
    float* Ks = Qs + S * HDP;
    float* Vs = Ks + S * HDP;
    float* Os = Vs + S * HDP;     // dO
    float* Pm = Os + S * HDP;     // [S][S+1] probabilities
    float* Dm = Pm + S * SP;      // [S][S+1] dS (already scaled)
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t ld = 3 * (int64_t)W;
    const T* base = qkv + (int64_t)b * S * ld + h * HD;
    T* dbase = d_qkv + (int64_t)b * S * ld + h * HD;
    load_head<T>(Qs, base, ld, S, tid);
    load_head<T>(Ks, base + W, ld, S, tid);
    load_head<T>(Vs, base + 2 * W, ld, S, tid);
    load_head<T>(Os, d_out + (int64_t)b * S * W + h * HD, W, S, tid);
    __syncthreads();
    const int iters = (S + 3) / 4;
    // phase A: per query row i: P, dS, dQ
    for (int it = 0; it < iters; ++it) {
        const int i = it * 4 + w;
        if (i < S) {
            float p0, p1;
            softmax_row<CAUSAL>(Qs, Ks, S, i, lane, scale, p0, p1);
            const int j0 = lane, j1 = 64 + lane;
            const float dp0 = (j0 < S) ? dot64(Os + i * HDP, Vs + j0 * HDP) : 0.f;
            const float dp1 = (j1 < S) ? dot64(Os + i * HDP, Vs + j1 * HDP) : 0.f;
            const float delta = wave_sum(p0 * dp0 + p1 * dp1);
            if (j0 < S) {
                Pm[i * SP + j0] = p0;
                Dm[i * SP + j0] = p0 * (dp0 - delta) * scale;
            }
            if (j1 < S) {
                Pm[i * SP + j1] = p1;
                Dm[i * SP + j1] = p1 * (dp1 - delta) * scale;
            }
        }
        __syncthreads();
        if (i < S) {
            const int jn = CAUSAL ? i + 1 : S;
            float dq = 0.f;
            for (int j = 0; j < jn; ++j) dq += Dm[i * SP + j] * Ks[j * HDP + lane];
            io<T>::st(dbase + (int64_t)i * ld + lane, dq);
        }
    }
    __syncthreads();
    // phase B: per key row j: dK, dV
    for (int j = w; j < S; j += 4) {
        const int i0 = CAUSAL ? j : 0;
        float dk = 0.f, dv = 0.f;
        for (int i = i0; i < S; ++i) {
            dk += Dm[i * SP + j] * Qs[i * HDP + lane];
            dv += Pm[i * SP + j] * Os[i * HDP + lane];
        }
        io<T>::st(dbase + (int64_t)j * ld + W + lane, dk);
        io<T>::st(dbase + (int64_t)j * ld + 2 * W + lane, dv);
    }
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
    if (bytes > 65536) {
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return sc_set_error((int)e, "attention: cannot reserve %zu bytes of LDS: %s", bytes, hipGetErrorString(e));
    }
    return SC_OK;
}

int check(const char* who, int dtype, int64_t batch, int64_t seq, int64_t width, int64_t heads) {
    SC_REQUIRE(dtype == SC_BF16 || dtype == SC_F32, SC_ERR_DTYPE, "%s: bad dtype %d", who, dtype);
    SC_REQUIRE(batch > 0 && seq > 0 && heads > 0 && width == heads * HD, SC_ERR_SHAPE, "%s: width %lld must be heads*64", who, (long long)width);
    SC_REQUIRE(seq <= (dtype == SC_BF16 ? 272 : 128), SC_ERR_SHAPE,
               "%s: sequence length %lld is not supported (bf16: <= 272, fp32: <= 128; longer fp32 sequences need a workspace and go through sc_block_fwd/bwd)",
               who, (long long)seq);
    SC_REQUIRE(batch * heads < (1ll << 31), SC_ERR_SHAPE, "%s: grid too large", who);
    return SC_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ fp32, 128 < seq <= 1024
// Parity path for long sequences (ViT-L/14: 257 tokens), one head at a time out of three pieces that exist already: the fp32
// MFMA GEMM on strided views of qkv (scores = Q K^T, O = P V, dV = P^T dO, dP = dO V^T, dQ = dS K, dK = dS^T Q), a row softmax
// and the dS row kernel below.  Needs 2 * seq^2 floats of workspace, which the public sc_attention_* entry points do not
// have: this path is reached through sc_block_fwd / sc_block_bwd (block.hip).  Launch bound by design: 3 launches per head
// forward, 8 backward.
int sc_gemm_f32_launch(int trans_a, int trans_b, int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, const float* b, int64_t ldb,
                       float* c, int64_t ldc, const EpiParams& epi, hipStream_t stream);

namespace {
// one wave per row: p[i][j] = softmax_j(scale * s[i][j]) over j < S (causal: j <= i), in place
__global__ __launch_bounds__(256) void softmax_rows_f32_kernel(float* p, int S, int64_t ld, float scale, int causal) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= S) return;
    float* row = p + (int64_t)i * ld;
    const int n = causal ? i + 1 : S;
    float m = -INFINITY;
    for (int j = lane; j < n; j += 64) m = fmaxf(m, row[j] * scale);
    m = wave_max(m);
    float l = 0.f;
    for (int j = lane; j < n; j += 64) l += expf(row[j] * scale - m);
    l = wave_sum(l);
    const float inv = 1.0f / l;
    for (int j = lane; j < S; j += 64) row[j] = j < n ? expf(row[j] * scale - m) * inv : 0.f;
}
// dS[i][j] = P[i][j] * (dP[i][j] - sum_j P[i][j] dP[i][j]) * scale, written over dP
__global__ __launch_bounds__(256) void attn_ds_rows_f32_kernel(const float* p, float* dp, int S, int64_t ld, float scale) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= S) return;
    const float* pr = p + (int64_t)i * ld;
    float* dr = dp + (int64_t)i * ld;
    float delta = 0.f;
    for (int j = lane; j < S; j += 64) delta += pr[j] * dr[j];
    delta = wave_sum(delta);
    for (int j = lane; j < S; j += 64) dr[j] = pr[j] * (dr[j] - delta) * scale;
}
}  // namespace

// fp32 backward kernel: q, k, v, dO tiles + scores + dP of one head in LDS
bool sc_attention_f32_bwd_fits_lds(int64_t seq) { return ((size_t)4 * seq * HDP + 2 * seq * (seq + 1)) * sizeof(float) <= 160 * 1024; }

int sc_attention_f32_composed_fwd(const float* qkv, float* out, int64_t batch, int64_t seq, int64_t width, int64_t heads, int causal, void* ws,
                                  size_t ws_bytes, hipStream_t st) {
    SC_REQUIRE(qkv && out && ws, SC_ERR_ARG, "attention (fp32, long): null argument");
    SC_REQUIRE(seq <= 1024 && width == heads * 64, SC_ERR_SHAPE, "attention (fp32, long): seq %lld > 1024 or width != heads*64", (long long)seq);
    const int64_t sp = (seq + 3) / 4 * 4;
    SC_REQUIRE(ws_bytes >= (size_t)seq * sp * sizeof(float) && sc_aligned(ws, 16), SC_ERR_WORKSPACE, "attention (fp32, long): workspace too small");
    float* P = (float*)ws;
    const int64_t ld = 3 * width;
    const EpiParams plain = epi_plain();
    for (int64_t b = 0; b < batch; ++b)
        for (int64_t h = 0; h < heads; ++h) {
            const float* q = qkv + b * seq * ld + h * 64;
            SC_TRY(sc_gemm_f32_launch(0, 1, seq, seq, 64, q, ld, q + width, ld, P, sp, plain, st));
            hipLaunchKernelGGL(softmax_rows_f32_kernel, dim3((unsigned)sc_cdiv(seq, 4)), dim3(256), 0, st, P, (int)seq, sp, 0.125f, causal);
            SC_TRY(sc_gemm_f32_launch(0, 0, seq, 64, seq, P, sp, q + 2 * width, ld, out + b * seq * width + h * 64, width, plain, st));
        }
    SC_CHECK_LAUNCH();
    return SC_OK;
}

int sc_attention_f32_composed_bwd(const float* qkv, const float* d_out, float* d_qkv, int64_t batch, int64_t seq, int64_t width, int64_t heads, int causal,
                                  void* ws, size_t ws_bytes, hipStream_t st) {
    SC_REQUIRE(qkv && d_out && d_qkv && ws, SC_ERR_ARG, "attention (fp32, long): null argument");
    SC_REQUIRE(seq <= 1024 && width == heads * 64, SC_ERR_SHAPE, "attention (fp32, long): seq %lld > 1024 or width != heads*64", (long long)seq);
    const int64_t sp = (seq + 3) / 4 * 4;
    SC_REQUIRE(ws_bytes >= (size_t)2 * seq * sp * sizeof(float) && sc_aligned(ws, 16), SC_ERR_WORKSPACE, "attention (fp32, long): workspace too small");
    float* P = (float*)ws;
    float* D = P + seq * sp;
    const int64_t ld = 3 * width;
    const EpiParams plain = epi_plain();
    for (int64_t b = 0; b < batch; ++b)
        for (int64_t h = 0; h < heads; ++h) {
            const float* q = qkv + b * seq * ld + h * 64;
            const float *k = q + width, *v = q + 2 * width;
            const float* go = d_out + b * seq * width + h * 64;
            float* dq = d_qkv + b * seq * ld + h * 64;
            SC_TRY(sc_gemm_f32_launch(0, 1, seq, seq, 64, q, ld, k, ld, P, sp, plain, st));                     // scores
            hipLaunchKernelGGL(softmax_rows_f32_kernel, dim3((unsigned)sc_cdiv(seq, 4)), dim3(256), 0, st, P, (int)seq, sp, 0.125f, causal);
            SC_TRY(sc_gemm_f32_launch(1, 0, seq, 64, seq, P, sp, go, width, dq + 2 * width, ld, plain, st));    // dV = P^T dO
            SC_TRY(sc_gemm_f32_launch(0, 1, seq, seq, 64, go, width, v, ld, D, sp, plain, st));                  // dP = dO V^T
            hipLaunchKernelGGL(attn_ds_rows_f32_kernel, dim3((unsigned)sc_cdiv(seq, 4)), dim3(256), 0, st, P, D, (int)seq, sp, 0.125f);
            SC_TRY(sc_gemm_f32_launch(0, 0, seq, 64, seq, D, sp, k, ld, dq, ld, plain, st));                     // dQ = dS K
            SC_TRY(sc_gemm_f32_launch(1, 0, seq, 64, seq, D, sp, q, ld, dq + width, ld, plain, st));            // dK = dS^T Q
        }
    SC_CHECK_LAUNCH();
    return SC_OK;
}

namespace {
int attention_fwd_impl(const void* qkv, void* out, float* lse, int dtype, int64_t batch, int64_t seq, int64_t width, int64_t heads, int causal, void* stream);
}
extern "C" int sc_attention_fwd(const void* qkv, void* out, int dtype, int64_t batch, int64_t seq, int64_t width, int64_t heads, int causal,
                                void* stream) {
    return attention_fwd_impl(qkv, out, nullptr, dtype, batch, seq, width, heads, causal, stream);
}
extern "C" int sc_attention_uses_stats(int dtype, int64_t seq) {
    return dtype == SC_BF16 && use_mfma() && !short_recompute() && sc_attention_long_uses_stats(seq) ? 1 : 0;
}
extern "C" int sc_attention_fwd_stats(const void* qkv, void* out, float* lse, int dtype, int64_t batch, int64_t seq, int64_t width, int64_t heads,
                                      int causal, void* stream) {
    return attention_fwd_impl(qkv, out, lse && sc_attention_uses_stats(dtype, seq) ? lse : nullptr, dtype, batch, seq, width, heads, causal, stream);
}
namespace {
int attention_fwd_impl(const void* qkv, void* out, float* lse, int dtype, int64_t batch, int64_t seq, int64_t width, int64_t heads, int causal, void* stream) {
    SC_TRY(check("sc_attention_fwd", dtype, batch, seq, width, heads));
    SC_REQUIRE(qkv && out, SC_ERR_ARG, "sc_attention_fwd: null argument");
    if (dtype == SC_BF16 && use_mfma()) {
        // forward, seq <= 80: one wave per head measured faster (81 / 86 us vs 85 / 112 us at the step's two shapes)
        int rc = short_recompute() ? 1 : sc_attention_mfma_fwd(qkv, out, batch, seq, width, heads, causal, (hipStream_t)stream);
        if (rc == 1) rc = sc_attention_long_fwd(qkv, out, batch, seq, width, heads, causal, (hipStream_t)stream, lse);
        if (rc != 1) return rc;
    }
    const size_t lds = ((size_t)3 * seq * HDP + 4 * 128) * sizeof(float);
    const float scale = 0.125f;  // 1/sqrt(64)
    const dim3 grid((unsigned)(batch * heads));
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_FWD(T, C)                                                                                          \
    do {                                                                                                          \
        SC_TRY(set_lds(attention_fwd_kernel<T, C>, lds));                                                         \
        hipLaunchKernelGGL((attention_fwd_kernel<T, C>), grid, dim3(256), lds, st, (const T*)qkv, (T*)out, (int)seq, (int)width, (int)heads, scale); \
    } while (0)
    if (dtype == SC_BF16) { if (causal) LAUNCH_FWD(bf16_t, true); else LAUNCH_FWD(bf16_t, false); }
    else { if (causal) LAUNCH_FWD(float, true); else LAUNCH_FWD(float, false); }
#undef LAUNCH_FWD
    SC_CHECK_LAUNCH();
    return SC_OK;
}
}  // namespace

namespace {
// cs_part != null: the MFMA kernels also write [batch][3 W] column sums of d_qkv per image; *cs_done tells whether they did
int attention_bwd_impl(const void* qkv, const void* d_out, void* d_qkv, int dtype, int64_t batch, int64_t seq, int64_t width, int64_t heads, int causal,
                       float* cs_part, bool* cs_done, void* stream, const void* fwd_out = nullptr, const float* lse = nullptr) {
    if (cs_done) *cs_done = false;
    SC_TRY(check("sc_attention_bwd", dtype, batch, seq, width, heads));
    SC_REQUIRE(qkv && d_out && d_qkv, SC_ERR_ARG, "sc_attention_bwd: null argument");
    if (dtype == SC_BF16 && use_mfma()) {
        // backward, seq <= 80: two waves per head on shared LDS images (208 us at S = 77, 242 us at S = 50; one wave per head: 286 / 303 us,
        // one workgroup per head: 341 / 262 us)
        int rc = short_recompute() ? 1 : sc_attention_mfma_bwd(qkv, d_out, d_qkv, batch, seq, width, heads, causal, cs_part, (hipStream_t)stream);
        if (rc == 1) rc = sc_attention_long_bwd(qkv, d_out, d_qkv, batch, seq, width, heads, causal, cs_part, (hipStream_t)stream, fwd_out, lse);
        if (rc != 1) {
            if (cs_done) *cs_done = rc == SC_OK && cs_part != nullptr;
            return rc;
        }
    }
    const size_t lds = ((size_t)4 * seq * HDP + 2 * seq * (seq + 1)) * sizeof(float);
    SC_REQUIRE(lds <= 160 * 1024, SC_ERR_SHAPE, "sc_attention_bwd: sequence length %lld needs %zu bytes of LDS (> 160 KiB)", (long long)seq, lds);
    const float scale = 0.125f;
    const dim3 grid((unsigned)(batch * heads));
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_BWD(T, C)                                                                                          \
    do {                                                                                                          \
        SC_TRY(set_lds(attention_bwd_kernel<T, C>, lds));                                                         \
        hipLaunchKernelGGL((attention_bwd_kernel<T, C>), grid, dim3(256), lds, st, (const T*)qkv, (const T*)d_out, (T*)d_qkv, (int)seq, (int)width, (int)heads, scale); \
    } while (0)
    if (dtype == SC_BF16) { if (causal) LAUNCH_BWD(bf16_t, true); else LAUNCH_BWD(bf16_t, false); }
    else { if (causal) LAUNCH_BWD(float, true); else LAUNCH_BWD(float, false); }
#undef LAUNCH_BWD
    SC_CHECK_LAUNCH();
    return SC_OK;
}
}  // namespace

extern "C" int sc_attention_bwd(const void* qkv, const void* d_out, void* d_qkv, int dtype, int64_t batch, int64_t seq, int64_t width,
                                int64_t heads, int causal, void* stream) {
    return attention_bwd_impl(qkv, d_out, d_qkv, dtype, batch, seq, width, heads, causal, nullptr, nullptr, stream);
}

extern "C" int sc_attention_bwd_stats(const void* qkv, const void* out, const float* lse, const void* d_out, void* d_qkv, int dtype, int64_t batch,
                                      int64_t seq, int64_t width, int64_t heads, int causal, float* colsum, int accumulate, void* ws, size_t ws_bytes,
                                      void* stream) {
    const bool stats = out && lse && sc_attention_uses_stats(dtype, seq);
    if (!colsum) return attention_bwd_impl(qkv, d_out, d_qkv, dtype, batch, seq, width, heads, causal, nullptr, nullptr, stream, stats ? out : nullptr, stats ? lse : nullptr);
    SC_REQUIRE(ws, SC_ERR_ARG, "sc_attention_bwd_stats: colsum needs a workspace");
    const int64_t n = 3 * width;
    float* part = (ws_bytes >= (size_t)batch * n * sizeof(float) && sc_aligned(ws, 16)) ? (float*)ws : nullptr;
    bool fused = false;
    SC_TRY(attention_bwd_impl(qkv, d_out, d_qkv, dtype, batch, seq, width, heads, causal, part, &fused, stream, stats ? out : nullptr, stats ? lse : nullptr));
    if (fused) return sc_colsum_reduce(part, (int)batch, n, colsum, accumulate, (hipStream_t)stream);
    return sc_colsum(d_qkv, dtype, batch * seq, n, n, colsum, accumulate, ws, ws_bytes, stream);
}

extern "C" int sc_attention_bwd_colsum(const void* qkv, const void* d_out, void* d_qkv, int dtype, int64_t batch, int64_t seq, int64_t width,
                                       int64_t heads, int causal, float* colsum, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    SC_REQUIRE(colsum && ws, SC_ERR_ARG, "sc_attention_bwd_colsum: null colsum / workspace");
    const int64_t n = 3 * width;
    float* part = (ws_bytes >= (size_t)batch * n * sizeof(float) && sc_aligned(ws, 16)) ? (float*)ws : nullptr;
    bool fused = false;
    SC_TRY(attention_bwd_impl(qkv, d_out, d_qkv, dtype, batch, seq, width, heads, causal, part, &fused, stream));
    if (fused) return sc_colsum_reduce(part, (int)batch, n, colsum, accumulate, (hipStream_t)stream);
    return sc_colsum(d_qkv, dtype, batch * seq, n, n, colsum, accumulate, ws, ws_bytes, stream);   // fp32 / VALU path: a pass over d_qkv
}
