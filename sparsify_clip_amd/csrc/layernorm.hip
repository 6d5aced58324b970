// LayerNorm (eps 1e-5, affine) forward/backward on the fp32 residual stream, and column sums (bias gradients).
// HBM-bound: one wave per row, 16-byte loads, fp32 statistics; the cross-row reductions (dgamma, dbeta, column
// sums) keep per-lane column accumulators in registers over a fixed set of rows per block, then a second kernel
// adds the per-block partials in block order - bit-stable, no float atomics.
#include "common.h"

namespace {

constexpr int LN_MAX_CHUNKS = 8;     // width <= 2048
constexpr int RED_MAX_BLOCKS = 1024; // partial slabs of the column reductions (4 workgroups per CU)

constexpr float LN_EPS = 1e-5f;

// RPW rows per wave: all loads of the wave's rows are issued before the first reduction (memory-level parallelism)
template <typename T, int RPW, int CH>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* x, int64_t rows, int width, const float* gamma, const float* beta,
                                                            T* y, float* mean_out, float* rstd_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    f32x4 v[RPW][CH];
    float s[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        s[r] = 0.f;
        const int64_t row = min(row0 + r, rows - 1);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int idx = c * 256 + lane * 4;
            if (idx < width) {
                v[r][c] = *(const f32x4*)(x + row * width + idx);
                s[r] += v[r][c][0] + v[r][c][1] + v[r][c][2] + v[r][c][3];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int64_t row = row0 + r;
        const float mean = wave_sum(s[r]) / (float)width;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int idx = c * 256 + lane * 4;
            if (idx < width) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = v[r][c][j] - mean;
                    q += d * d;
                }
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)width + LN_EPS);
        if (row >= rows) continue;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int idx = c * 256 + lane * 4;
            if (idx < width) {
                const f32x4 g = *(const f32x4*)(gamma + idx), b = *(const f32x4*)(beta + idx);
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (v[r][c][j] - mean) * rstd * g[j] + b[j];
                io<T>::st4(y + row * width + idx, o);
            }
        }
        if (lane == 0) {
            mean_out[row] = mean;
            rstd_out[row] = rstd;
        }
    }
}

// dx = dres + rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy * gamma
// partial[block][0][w] = sum_rows dy*xhat ; partial[block][1][w] = sum_rows dy ; partial[block][2][w] = sum_rows dx (NSUM == 3)
template <typename T, int NSUM, int CH>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                                            int64_t rows, int width, const float* dres, float* dx, T* dx_cast, float* partial) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [wave][which][width]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f32x4 ag[CH], ab[CH], ad[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) ag[c] = ab[c] = ad[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float invw = 1.0f / (float)width;
    for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < rows; row += (int64_t)gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        f32x4 xh[CH], g[CH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int idx = c * 256 + lane * 4;
            if (idx < width) {
                const f32x4 xv = *(const f32x4*)(x + row * width + idx);
                const f32x4 d = io<T>::ld4(dy + row * width + idx);
                const f32x4 gm = *(const f32x4*)(gamma + idx);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xh[c][j] = (xv[j] - mu) * rs;
                    g[c][j] = d[j] * gm[j];
                    s1 += g[c][j];
                    s2 += g[c][j] * xh[c][j];
                    ag[c][j] += d[j] * xh[c][j];
                    ab[c][j] += d[j];
                }
            }
        }
        s1 = wave_sum(s1) * invw;
        s2 = wave_sum(s2) * invw;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int idx = c * 256 + lane * 4;
            if (idx < width) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = rs * (g[c][j] - s1 - xh[c][j] * s2);
                if (dres) o += *(const f32x4*)(dres + row * width + idx);
                if (NSUM == 3) ad[c] += o;
                *(f32x4*)(dx + row * width + idx) = o;
                if (dx_cast) io<T>::st4(dx_cast + row * width + idx, o);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int idx = c * 256 + lane * 4;
        if (idx < width) {
            *(f32x4*)&red[(w * NSUM + 0) * width + idx] = ag[c];
            *(f32x4*)&red[(w * NSUM + 1) * width + idx] = ab[c];
            if (NSUM == 3) *(f32x4*)&red[(w * NSUM + 2) * width + idx] = ad[c];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NSUM * width; i += 256) {
        const int which = i / width, col = i - which * width;
        float s = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) s += red[(ww * NSUM + which) * width + col];
        partial[((int64_t)blockIdx.x * NSUM + which) * width + col] = s;
    }
}

// out[which][col] (+)= sum_b partial[b][which][col]     (nwhich slabs of `width` columns per block; width % 4 == 0)
// A workgroup owns 32 consecutive columns of the flattened [nwhich * width] row: 8 threads x 16 bytes across, 32 row groups down;
// every load is a 16-byte piece of a 128-byte run, the 32 group sums are combined in a fixed order.
__device__ __forceinline__ void partial_reduce_body(const float* partial, int nblocks, int nwhich, int width, float* out0, float* out1, float* out2,
                                                    int accumulate, int block) {
    __shared__ f32x4 sm[32][9];
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    const int total = nwhich * width;
    const int i = (block * 8 + tx) * 4;
    const bool live = i < total;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const float* p = partial + i;
        int b = ty;
        for (; b + 96 < nblocks; b += 128) {   // four independent loads in flight per thread
            const f32x4 v0 = *(const f32x4*)(p + (int64_t)b * total), v1 = *(const f32x4*)(p + (int64_t)(b + 32) * total);
            const f32x4 v2 = *(const f32x4*)(p + (int64_t)(b + 64) * total), v3 = *(const f32x4*)(p + (int64_t)(b + 96) * total);
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; b < nblocks; b += 32) s += *(const f32x4*)(p + (int64_t)b * total);
    }
    sm[ty][tx] = s;
    __syncthreads();
    if (ty != 0 || !live) return;
    s = sm[0][tx];
#pragma unroll
    for (int k = 1; k < 32; ++k) s += sm[k][tx];
    const int which = i / width, col = i - which * width;
    float* out = which == 0 ? out0 : (which == 1 ? out1 : out2);
    if (!out) return;
    if ((((size_t)(out + col)) & 15) == 0) {
        f32x4* o = (f32x4*)(out + col);
        *o = accumulate ? *o + s : s;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) out[col + e] = accumulate ? out[col + e] + s[e] : s[e];
    }
}
__global__ __launch_bounds__(256) void partial_reduce_kernel(const float* partial, int nblocks, int nwhich, int width, float* out0, float* out1,
                                                             float* out2, int accumulate) {
    partial_reduce_body(partial, nblocks, nwhich, width, out0, out1, out2, accumulate, (int)blockIdx.x);
}
// several reductions in ONE launch (sc_reduce_defer_*): workgroup b belongs to the job whose range of blocks holds it; the arithmetic of a job
// is partial_reduce_kernel's, bit for bit
__global__ __launch_bounds__(256) void partial_reduce_jobs_kernel(ScReduceJobs jobs) {
    int j = 0;
#pragma unroll
    for (int k = 1; k < SC_REDUCE_JOBS; ++k)
        if (k < jobs.n && (int)blockIdx.x >= jobs.job[k].first_block) j = k;
    const ScReduceJob& q = jobs.job[j];
    partial_reduce_body(q.partial, q.nblocks, q.nwhich, q.width, q.out0, q.out1, q.out2, q.accumulate, (int)blockIdx.x - q.first_block);
}

// column sums of x [rows, n]: thread = 4 consecutive columns, blockIdx.y = row slab
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* x, int64_t rows, int n, int64_t ld, int64_t rows_per_block, float* partial) {
    const int col = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (col >= n) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int64_t r = r0; r < r1; ++r) s += io<T>::ld4(x + r * ld + col);
    *(f32x4*)(partial + (int64_t)blockIdx.y * n + col) = s;
}

// act = GELU(pre), bf16 -> bf16, 8 elements (16 bytes) per thread (un-fused form of the c_fc epilogue, SC_BLOCK_UNFUSE_GELU=1)
__global__ __launch_bounds__(256) void gelu_fwd_bf16_kernel(const bf16_t* pre, bf16_t* act, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const uint4 u = *(const uint4*)(pre + i * 8);
    const f32x2 a = gelu_fast2(f32x2{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u)});
    const f32x2 b = gelu_fast2(f32x2{__uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)});
    const f32x2 c = gelu_fast2(f32x2{__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u)});
    const f32x2 d = gelu_fast2(f32x2{__uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u)});
    uint4 o;
    o.x = pack2_bf16(a[0], a[1]);
    o.y = pack2_bf16(b[0], b[1]);
    o.z = pack2_bf16(c[0], c[1]);
    o.w = pack2_bf16(d[0], d[1]);
    *(uint4*)(act + i * 8) = o;
}
// dh <- dh * GELU'(pre) in place (bf16), column sums of the result per row slab -> partial [nslab][n]
__global__ __launch_bounds__(256) void dgelu_mul_colsum_bf16_kernel(bf16_t* dh, const bf16_t* pre, int64_t rows, int n, int64_t rows_per_block, float* partial) {
    const int col = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (col >= n) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int64_t r = r0; r < r1; ++r) {
        const uint2 g = *(const uint2*)(dh + r * n + col), h = *(const uint2*)(pre + r * n + col);
        const f32x4 v = gelu_grad_mul4(f32x4{__uint_as_float(g.x << 16), __uint_as_float(g.x & 0xffff0000u), __uint_as_float(g.y << 16),
                                             __uint_as_float(g.y & 0xffff0000u)}, h.x, h.y);
        uint2 o;
        o.x = pack2_bf16(v[0], v[1]);
        o.y = pack2_bf16(v[2], v[3]);
        *(uint2*)(dh + r * n + col) = o;
        s[0] += __uint_as_float(o.x << 16); s[1] += __uint_as_float(o.x & 0xffff0000u);
        s[2] += __uint_as_float(o.y << 16); s[3] += __uint_as_float(o.y & 0xffff0000u);
    }
    *(f32x4*)(partial + (int64_t)blockIdx.y * n + col) = s;
}

}  // namespace

// see common.h (ScReduceJobs)
namespace { thread_local ScReduceJobs* g_reduce_defer = nullptr; }
void sc_reduce_defer_begin(ScReduceJobs* jobs) { jobs->n = 0; g_reduce_defer = jobs; }
void sc_reduce_defer_cancel() { g_reduce_defer = nullptr; }
int sc_reduce_defer_flush(hipStream_t st) {
    ScReduceJobs* l = g_reduce_defer;
    g_reduce_defer = nullptr;
    if (!l || l->n == 0) return SC_OK;
    int blocks = 0;
    for (int k = 0; k < l->n; ++k) {
        l->job[k].first_block = blocks;
        blocks += (int)sc_cdiv((int64_t)l->job[k].nwhich * l->job[k].width, 32);
    }
    hipLaunchKernelGGL(partial_reduce_jobs_kernel, dim3((unsigned)blocks), dim3(256), 0, st, *l);
    SC_CHECK_LAUNCH();
    return SC_OK;
}
// the second stage of every column reduction of this file: launched now, or recorded when the calling thread is deferring
static int launch_partial_reduce(const float* partial, int nblocks, int nwhich, int width, float* out0, float* out1, float* out2, int accumulate, hipStream_t st) {
    ScReduceJobs* l = g_reduce_defer;
    if (l && l->n < SC_REDUCE_JOBS) {
        ScReduceJob& q = l->job[l->n++];
        q.partial = partial; q.nblocks = nblocks; q.nwhich = nwhich; q.width = width; q.accumulate = accumulate; q.first_block = 0;
        q.out0 = out0; q.out1 = out1; q.out2 = out2;
        return SC_OK;
    }
    hipLaunchKernelGGL(partial_reduce_kernel, dim3((unsigned)sc_cdiv((int64_t)nwhich * width, 32)), dim3(256), 0, st, partial, nblocks, nwhich, width, out0, out1, out2,
                       accumulate);
    SC_CHECK_LAUNCH();
    return SC_OK;
}

// un-fused GELU pieces (A/B experiment against the GEMM epilogues; bf16 only)
int sc_gelu_fwd_bf16(const void* pre, void* act, int64_t n_elems, hipStream_t st) {
    SC_REQUIRE(pre && act && n_elems > 0 && n_elems % 8 == 0, SC_ERR_ARG, "sc_gelu_fwd_bf16: bad argument");
    hipLaunchKernelGGL(gelu_fwd_bf16_kernel, dim3((unsigned)sc_cdiv(n_elems / 8, 256)), dim3(256), 0, st, (const bf16_t*)pre, (bf16_t*)act, n_elems / 8);
    SC_CHECK_LAUNCH();
    return SC_OK;
}
int sc_dgelu_mul_colsum_bf16(void* dh, const void* pre, int64_t rows, int64_t n, float* colsum, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
    SC_REQUIRE(dh && pre && colsum && ws && rows > 0 && n > 0 && n % 4 == 0, SC_ERR_ARG, "sc_dgelu_mul_colsum_bf16: bad argument");
    const int slabs = (int)min((int64_t)(n <= 1024 ? RED_MAX_BLOCKS : RED_MAX_BLOCKS / 3), sc_cdiv(rows, 32));
    SC_REQUIRE(ws_bytes >= (size_t)slabs * n * sizeof(float), SC_ERR_WORKSPACE, "sc_dgelu_mul_colsum_bf16: workspace too small");
    const int64_t rpb = sc_cdiv(rows, slabs);
    const int nslab = (int)sc_cdiv(rows, rpb);
    hipLaunchKernelGGL(dgelu_mul_colsum_bf16_kernel, dim3((unsigned)sc_cdiv(n, 1024), nslab), dim3(256), 0, st, (bf16_t*)dh, (const bf16_t*)pre, rows, (int)n, rpb,
                       (float*)ws);
    SC_CHECK_LAUNCH();
    return launch_partial_reduce((const float*)ws, nslab, 1, (int)n, colsum, nullptr, nullptr, accumulate, st);
}

namespace {
}  // namespace

extern "C" int sc_layernorm_fwd(const float* x, int64_t rows, int64_t width, const float* gamma, const float* beta, void* y, int dtype,
                                float* mean, float* rstd, void* stream) {
    SC_REQUIRE(x && gamma && beta && y && mean && rstd, SC_ERR_ARG, "sc_layernorm_fwd: null argument");
    SC_REQUIRE(rows > 0 && width > 0 && width % 4 == 0 && width <= LN_MAX_CHUNKS * 256, SC_ERR_SHAPE,
               "sc_layernorm_fwd: width %lld must be a multiple of 4 and <= %d", (long long)width, LN_MAX_CHUNKS * 256);
    SC_REQUIRE(sc_aligned(x, 16) && sc_aligned(y, 8) && sc_aligned(gamma, 16) && sc_aligned(beta, 16), SC_ERR_ALIGN, "sc_layernorm_fwd: misaligned");
    // the chunk count (256 columns per chunk) is a compile-time parameter: registers, and with them the waves in flight,
    // follow the actual width instead of the 2048-column maximum; narrow rows take 4 rows per wave
    const int ch = width <= 512 ? 2 : width <= 768 ? 3 : width <= 1024 ? 4 : 8;
    const bool multi = ch <= 4;
    // rows per wave: 1 (rounds 1-2 took 4 for widths <= 1024 - more loads in flight per wave, but 64 data registers at width 1024 and fewer
    // waves per SIMD; measured forward, 4 / 2 / 1 rows per wave: 48.9 / 42.3 / 37.8 us at [51200 x 768], 204.3 / 158.8 / 152.1 us at
    // [131584 x 1024], profiles/r03_layernorm_times_v2.txt); SC_LN_FWD_RPW (debug builds) selects 4 or 2 for A/B runs
    static const int rpw_env = [] { const char* e = sc_debug_env("SC_LN_FWD_RPW"); return e ? atoi(e) : 1; }();
    const int rpw = multi && (rpw_env == 4 || rpw_env == 2) ? rpw_env : 1;
    const dim3 grid((unsigned)sc_cdiv(rows, 4 * rpw));
    hipStream_t st = (hipStream_t)stream;
    if (dtype != SC_BF16 && dtype != SC_F32) return sc_set_error(SC_ERR_DTYPE, "sc_layernorm_fwd: bad dtype %d", dtype);
#define LN_FWD(T, R, C) hipLaunchKernelGGL((layernorm_fwd_kernel<T, R, C>), grid, dim3(256), 0, st, x, rows, (int)width, gamma, beta, (T*)y, mean, rstd)
#define LN_FWD_R(T, C) do { if (rpw == 4) LN_FWD(T, 4, C); else if (rpw == 2) LN_FWD(T, 2, C); else LN_FWD(T, 1, C); } while (0)
#define LN_FWD_T(T) do { if (ch == 2) LN_FWD_R(T, 2); else if (ch == 3) LN_FWD_R(T, 3); else if (ch == 4) LN_FWD_R(T, 4); else LN_FWD(T, 1, 8); } while (0)
    if (dtype == SC_BF16) LN_FWD_T(bf16_t); else LN_FWD_T(float);
#undef LN_FWD_T
#undef LN_FWD_R
#undef LN_FWD
    SC_CHECK_LAUNCH();
    return SC_OK;
}

// workgroups of layernorm_bwd_kernel<T, NSUM, CH> the device holds at once (occupancy query per instantiation and LDS size, cached)
namespace {
template <typename T, int NS, int C>
int ln_bwd_resident_of(size_t lds_bytes) {
    static size_t seen_lds = (size_t)-1;
    static int seen = 0;
    if (seen_lds != lds_bytes) {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, layernorm_bwd_kernel<T, NS, C>, 256, lds_bytes) != hipSuccess || per_cu < 1) per_cu = 1;
        seen = per_cu * cus;
        seen_lds = lds_bytes;
    }
    return seen;
}
int ln_bwd_resident(int dtype, int nsum, int ch, size_t lds_bytes) {
#define LNR(T, NS) (ch == 2 ? ln_bwd_resident_of<T, NS, 2>(lds_bytes) : ch == 3 ? ln_bwd_resident_of<T, NS, 3>(lds_bytes) : ch == 4 ? ln_bwd_resident_of<T, NS, 4>(lds_bytes) : ln_bwd_resident_of<T, NS, 8>(lds_bytes))
    if (dtype == SC_BF16) return nsum == 3 ? LNR(bf16_t, 3) : LNR(bf16_t, 2);
    return nsum == 3 ? LNR(float, 3) : LNR(float, 2);
#undef LNR
}
}  // namespace

extern "C" int sc_layernorm_bwd(const void* dy, int dtype, const float* x, const float* mean, const float* rstd, const float* gamma, int64_t rows,
                                int64_t width, const float* dres, float* dx, void* dx_cast, float* dgamma, float* dbeta, float* dx_colsum,
                                int accumulate, void* ws, size_t ws_bytes, void* stream) {
    SC_REQUIRE(dy && x && mean && rstd && gamma && dx, SC_ERR_ARG, "sc_layernorm_bwd: null argument");
    SC_REQUIRE(rows > 0 && width > 0 && width % 4 == 0 && width <= LN_MAX_CHUNKS * 256, SC_ERR_SHAPE, "sc_layernorm_bwd: bad width %lld", (long long)width);
    const int nsum = dx_colsum ? 3 : 2;
    const size_t lds_bytes = (size_t)4 * nsum * width * sizeof(float);
    const int ch = width <= 512 ? 2 : width <= 768 ? 3 : width <= 1024 ? 4 : 8;
    // One ROUND of workgroups: the rows are dealt to the blocks statically, so a grid larger than what the chip holds at once ends on a
    // partly filled second round (width 1024: 150 VGPRs = 3 blocks per CU, 1024 blocks = 1.33 rounds - the ViT-L/14 LayerNorm backward ran
    // at 3.9 TB/s of its bytes where the narrower ones reach 5.1-5.3)
    const int resident = ln_bwd_resident(dtype, nsum, ch, lds_bytes);
    const int nblocks = (int)min((int64_t)min(RED_MAX_BLOCKS, resident), sc_cdiv(rows, 4));
    SC_REQUIRE(ws && ws_bytes >= (size_t)nblocks * nsum * width * sizeof(float), SC_ERR_WORKSPACE, "sc_layernorm_bwd: workspace too small");
    SC_REQUIRE(sc_aligned(ws, 16) && sc_aligned(x, 16) && sc_aligned(dx, 16) && sc_aligned(dy, 8), SC_ERR_ALIGN, "sc_layernorm_bwd: misaligned");
    hipStream_t st = (hipStream_t)stream;
    if (dtype != SC_BF16 && dtype != SC_F32) return sc_set_error(SC_ERR_DTYPE, "sc_layernorm_bwd: bad dtype %d", dtype);
#define LN_BWD(T, NS, C) hipLaunchKernelGGL((layernorm_bwd_kernel<T, NS, C>), dim3(nblocks), dim3(256), lds_bytes, st, (const T*)dy, x, mean, rstd, gamma, rows, (int)width, dres, dx, (T*)dx_cast, (float*)ws)
#define LN_BWD_C(T, NS) do { if (ch == 2) LN_BWD(T, NS, 2); else if (ch == 3) LN_BWD(T, NS, 3); else if (ch == 4) LN_BWD(T, NS, 4); else LN_BWD(T, NS, 8); } while (0)
    if (dtype == SC_BF16) { if (nsum == 3) LN_BWD_C(bf16_t, 3); else LN_BWD_C(bf16_t, 2); }
    else { if (nsum == 3) LN_BWD_C(float, 3); else LN_BWD_C(float, 2); }
#undef LN_BWD_C
#undef LN_BWD
    SC_CHECK_LAUNCH();
    if (dgamma || dbeta || dx_colsum) return launch_partial_reduce((const float*)ws, nblocks, nsum, (int)width, dgamma, dbeta, dx_colsum, accumulate, st);
    return SC_OK;
}

// out[n] (+)= sum over nslab partial rows [nslab][n] (fixed order); second stage of sc_colsum, also used by the GEMM's fused column sums
int sc_colsum_reduce(const float* partial, int nslab, int64_t n, float* out, int accumulate, hipStream_t st) {
    return launch_partial_reduce(partial, nslab, 1, (int)n, out, nullptr, nullptr, accumulate, st);
}

extern "C" int sc_colsum(const void* x, int dtype, int64_t rows, int64_t n, int64_t ld, float* out, int accumulate, void* ws, size_t ws_bytes,
                         void* stream) {
    SC_REQUIRE(x && out && rows > 0 && n > 0, SC_ERR_ARG, "sc_colsum: bad argument");
    SC_REQUIRE(n % 4 == 0 && ld % 4 == 0 && ld >= n, SC_ERR_SHAPE, "sc_colsum: n and ld must be multiples of 4");
    const int slabs = (int)min((int64_t)(n <= 1024 ? RED_MAX_BLOCKS : RED_MAX_BLOCKS / 3), sc_cdiv(rows, 32));
    SC_REQUIRE(ws && ws_bytes >= (size_t)slabs * n * sizeof(float), SC_ERR_WORKSPACE, "sc_colsum: workspace too small");
    const int64_t rpb = sc_cdiv(rows, slabs);
    const int nslab = (int)sc_cdiv(rows, rpb);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)sc_cdiv(n, 1024), nslab);
    if (dtype == SC_BF16)
        hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, rows, (int)n, ld, rpb, (float*)ws);
    else if (dtype == SC_F32)
        hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, st, (const float*)x, rows, (int)n, ld, rpb, (float*)ws);
    else
        return sc_set_error(SC_ERR_DTYPE, "sc_colsum: bad dtype %d", dtype);
    SC_CHECK_LAUNCH();
    return launch_partial_reduce((const float*)ws, nslab, 1, (int)n, out, nullptr, nullptr, accumulate, st);
}
