// LayerNorm (eps 1e-5, affine) forward/backward on the fp32 residual stream, and column sums (bias gradients).
// HBM-bound: one wave per row, 16-byte loads, fp32 statistics; the cross-row reductions (dgamma, dbeta, column
// sums) keep per-lane column accumulators in registers over a fixed set of rows per block, then a second kernel
// adds the per-block partials in block order - bit-stable, no float atomics.
#include "common.h"

namespace {

constexpr int LN_MAX_CHUNKS = 8;     // width <= 2048
constexpr int RED_MAX_BLOCKS = 256;  // partial slabs of the column reductions
constexpr float LN_EPS = 1e-5f;

template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* x, int64_t rows, int width, const float* gamma, const float* beta,
                                                            T* y, float* mean_out, float* rstd_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * width;
    f32x4 v[LN_MAX_CHUNKS];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int idx = c * 256 + lane * 4;
        if (idx < width) {
            v[c] = *(const f32x4*)(xr + idx);
            s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
        }
    }
    const float mean = wave_sum(s) / (float)width;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int idx = c * 256 + lane * 4;
        if (idx < width) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[c][j] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)width + LN_EPS);
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int idx = c * 256 + lane * 4;
        if (idx < width) {
            const f32x4 g = *(const f32x4*)(gamma + idx), b = *(const f32x4*)(beta + idx);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[c][j] - mean) * rstd * g[j] + b[j];
            io<T>::st4(y + row * width + idx, o);
        }
    }
    if (lane == 0) {
        mean_out[row] = mean;
        rstd_out[row] = rstd;
    }
}

// dx = dres + rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy * gamma
// partial[block][0][w] = sum_rows dy*xhat ; partial[block][1][w] = sum_rows dy
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                                            int64_t rows, int width, const float* dres, float* dx, T* dx_cast, float* partial) {
    __shared__ float red[4 * 2 * LN_MAX_CHUNKS * 256];   // [wave][which][width<=2048]  (64 KiB)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f32x4 ag[LN_MAX_CHUNKS], ab[LN_MAX_CHUNKS];
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) ag[c] = ab[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float invw = 1.0f / (float)width;
    for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < rows; row += (int64_t)gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        f32x4 xh[LN_MAX_CHUNKS], g[LN_MAX_CHUNKS];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
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
        for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
            const int idx = c * 256 + lane * 4;
            if (idx < width) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = rs * (g[c][j] - s1 - xh[c][j] * s2);
                if (dres) o += *(const f32x4*)(dres + row * width + idx);
                *(f32x4*)(dx + row * width + idx) = o;
                if (dx_cast) io<T>::st4(dx_cast + row * width + idx, o);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int idx = c * 256 + lane * 4;
        if (idx < width) {
            *(f32x4*)&red[(w * 2 + 0) * (LN_MAX_CHUNKS * 256) + idx] = ag[c];
            *(f32x4*)&red[(w * 2 + 1) * (LN_MAX_CHUNKS * 256) + idx] = ab[c];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * width; i += 256) {
        const int which = i / width, col = i - which * width;
        float s = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) s += red[(ww * 2 + which) * (LN_MAX_CHUNKS * 256) + col];
        partial[((int64_t)blockIdx.x * 2 + which) * width + col] = s;
    }
}

// out[which][col] (+)= sum_b partial[b][which][col]     (nwhich slabs of `width` columns per block)
__global__ __launch_bounds__(256) void partial_reduce_kernel(const float* partial, int nblocks, int nwhich, int width, float* out0, float* out1,
                                                             int accumulate) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nwhich * width) return;
    const int which = i / width, col = i - which * width;
    float s = 0.f;
    for (int b = 0; b < nblocks; ++b) s += partial[((int64_t)b * nwhich + which) * width + col];
    float* out = which == 0 ? out0 : out1;
    if (!out) return;
    out[col] = accumulate ? out[col] + s : s;
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

}  // namespace

extern "C" int sc_layernorm_fwd(const float* x, int64_t rows, int64_t width, const float* gamma, const float* beta, void* y, int dtype,
                                float* mean, float* rstd, void* stream) {
    SC_REQUIRE(x && gamma && beta && y && mean && rstd, SC_ERR_ARG, "sc_layernorm_fwd: null argument");
    SC_REQUIRE(rows > 0 && width > 0 && width % 4 == 0 && width <= LN_MAX_CHUNKS * 256, SC_ERR_SHAPE,
               "sc_layernorm_fwd: width %lld must be a multiple of 4 and <= %d", (long long)width, LN_MAX_CHUNKS * 256);
    SC_REQUIRE(sc_aligned(x, 16) && sc_aligned(y, 8) && sc_aligned(gamma, 16) && sc_aligned(beta, 16), SC_ERR_ALIGN, "sc_layernorm_fwd: misaligned");
    const dim3 grid((unsigned)sc_cdiv(rows, 4));
    if (dtype == SC_BF16)
        hipLaunchKernelGGL(layernorm_fwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, x, rows, (int)width, gamma, beta, (bf16_t*)y, mean, rstd);
    else if (dtype == SC_F32)
        hipLaunchKernelGGL(layernorm_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, x, rows, (int)width, gamma, beta, (float*)y, mean, rstd);
    else
        return sc_set_error(SC_ERR_DTYPE, "sc_layernorm_fwd: bad dtype %d", dtype);
    SC_CHECK_LAUNCH();
    return SC_OK;
}

extern "C" int sc_layernorm_bwd(const void* dy, int dtype, const float* x, const float* mean, const float* rstd, const float* gamma, int64_t rows,
                                int64_t width, const float* dres, float* dx, void* dx_cast, float* dgamma, float* dbeta, int accumulate,
                                void* ws, size_t ws_bytes, void* stream) {
    SC_REQUIRE(dy && x && mean && rstd && gamma && dx, SC_ERR_ARG, "sc_layernorm_bwd: null argument");
    SC_REQUIRE(rows > 0 && width > 0 && width % 4 == 0 && width <= LN_MAX_CHUNKS * 256, SC_ERR_SHAPE, "sc_layernorm_bwd: bad width %lld", (long long)width);
    const int nblocks = (int)min((int64_t)RED_MAX_BLOCKS, sc_cdiv(rows, 4));
    SC_REQUIRE(ws && ws_bytes >= (size_t)nblocks * 2 * width * sizeof(float), SC_ERR_WORKSPACE, "sc_layernorm_bwd: workspace too small");
    SC_REQUIRE(sc_aligned(ws, 16) && sc_aligned(x, 16) && sc_aligned(dx, 16) && sc_aligned(dy, 8), SC_ERR_ALIGN, "sc_layernorm_bwd: misaligned");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SC_BF16)
        hipLaunchKernelGGL(layernorm_bwd_kernel<bf16_t>, dim3(nblocks), dim3(256), 0, st, (const bf16_t*)dy, x, mean, rstd, gamma, rows, (int)width, dres, dx, (bf16_t*)dx_cast, (float*)ws);
    else if (dtype == SC_F32)
        hipLaunchKernelGGL(layernorm_bwd_kernel<float>, dim3(nblocks), dim3(256), 0, st, (const float*)dy, x, mean, rstd, gamma, rows, (int)width, dres, dx, (float*)dx_cast, (float*)ws);
    else
        return sc_set_error(SC_ERR_DTYPE, "sc_layernorm_bwd: bad dtype %d", dtype);
    if (dgamma || dbeta)
        hipLaunchKernelGGL(partial_reduce_kernel, dim3((unsigned)sc_cdiv(2 * width, 256)), dim3(256), 0, st, (const float*)ws, nblocks, 2, (int)width, dgamma,
                           dbeta, accumulate);
    SC_CHECK_LAUNCH();
    return SC_OK;
}

extern "C" int sc_colsum(const void* x, int dtype, int64_t rows, int64_t n, int64_t ld, float* out, int accumulate, void* ws, size_t ws_bytes,
                         void* stream) {
    SC_REQUIRE(x && out && rows > 0 && n > 0, SC_ERR_ARG, "sc_colsum: bad argument");
    SC_REQUIRE(n % 4 == 0 && ld % 4 == 0 && ld >= n, SC_ERR_SHAPE, "sc_colsum: n and ld must be multiples of 4");
    const int slabs = (int)min((int64_t)RED_MAX_BLOCKS, sc_cdiv(rows, 32));
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
    hipLaunchKernelGGL(partial_reduce_kernel, dim3((unsigned)sc_cdiv(n, 256)), dim3(256), 0, st, (const float*)ws, nslab, 1, (int)n, out, (float*)nullptr,
                       accumulate);
    SC_CHECK_LAUNCH();
    return SC_OK;
}
