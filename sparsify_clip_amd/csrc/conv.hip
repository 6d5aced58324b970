// ModifiedResNet (open_clip "RN50", the `model:` of every reference YAML) pieces that are not GEMMs: 3x3 convolution lowering,
// anti-aliasing average pools, BatchNorm in training mode (batch statistics, running-statistics update, fused ReLU / residual join)
// and the token assembly of the attention pool.  Activations are NHWC, i.e. row-major [B*H*W, C] matrices, so every 1x1
// convolution is the NT GEMM of gemm_bf16.hip / gemm_f32.hip as it stands and a 3x3 convolution is im2col + the same GEMM.
// All kernels here are HBM-bound streaming kernels: coalesced along C, fp32 arithmetic, fixed-order reductions (no atomics).
#include "common.h"

namespace {

constexpr int BN_MAX_BLOCKS = 1024;

// ---------------------------------------------------------------- 3x3 convolution lowering (padding 1, stride 1 or 2)
// out[(b, yo, xo)][tap * C + c] = x[b][yo*stride + ky - 1][xo*stride + kx - 1][c]  (tap = 3 ky + kx; 0 outside; columns >= 9C are 0)
// NCHW = true reads an fp32 image tensor [B, C, H, W] (the stem's first convolution) instead of an NHWC activation.
template <typename TI, typename TO, bool NCHW>
__global__ __launch_bounds__(256) void im2col3x3_kernel(const TI* x, int B, int H, int W, int C, int stride, int Ho, int Wo, int kpad, TO* out) {
    const int64_t total = (int64_t)B * Ho * Wo * kpad;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int col = (int)(i % kpad);
        const int64_t row = i / kpad;
        float v = 0.f;
        if (col < 9 * C) {
            const int tap = col / C, c = col - tap * C;
            const int xo = (int)(row % Wo), yo = (int)((row / Wo) % Ho), b = (int)(row / ((int64_t)Wo * Ho));
            const int y = yo * stride + tap / 3 - 1, xx = xo * stride + tap % 3 - 1;
            if (y >= 0 && y < H && xx >= 0 && xx < W)
                v = NCHW ? io<TI>::ld(x + (((int64_t)b * C + c) * H + y) * W + xx) : io<TI>::ld(x + (((int64_t)b * H + y) * W + xx) * C + c);
        }
        io<TO>::st(out + i, v);
    }
}
// dx[b][y][x][c] = sum over the taps that read this pixel of dcols[(b, yo, xo)][tap * C + c]   (gather form: fixed order)
template <typename T>
__global__ __launch_bounds__(256) void col2im3x3_kernel(const T* dcols, int B, int H, int W, int C, int stride, int Ho, int Wo, int kpad, T* dx) {
    const int64_t total = (int64_t)B * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t p = i / C;
        const int xx = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((int64_t)W * H));
        float s = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = y + 1 - ky;
            if (ty < 0 || ty % stride != 0 || ty / stride >= Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = xx + 1 - kx;
                if (tx < 0 || tx % stride != 0 || tx / stride >= Wo) continue;
                s += io<T>::ld(dcols + (((int64_t)b * Ho + ty / stride) * Wo + tx / stride) * kpad + (3 * ky + kx) * C + c);
            }
        }
        io<T>::st(dx + i, s);
    }
}

// ---------------------------------------------------------------- AvgPool2d(k), k = stride (2 in the network)
template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* x, int B, int H, int W, int C, int k, T* y) {
    const int Ho = H / k, Wo = W / k;
    const int64_t total = (int64_t)B * Ho * Wo * C;
    const float inv = 1.0f / (float)(k * k);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t p = i / C;
        const int xo = (int)(p % Wo), yo = (int)((p / Wo) % Ho), b = (int)(p / ((int64_t)Wo * Ho));
        float s = 0.f;
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx) s += io<T>::ld(x + (((int64_t)b * H + yo * k + dy) * W + xo * k + dx) * C + c);
        io<T>::st(y + i, s * inv);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* dy, int B, int H, int W, int C, int k, T* dx) {
    const int Ho = H / k, Wo = W / k;
    const int64_t total = (int64_t)B * H * W * C;
    const float inv = 1.0f / (float)(k * k);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t p = i / C;
        const int xx = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((int64_t)W * H));
        float v = 0.f;
        if (y / k < Ho && xx / k < Wo) v = io<T>::ld(dy + (((int64_t)b * Ho + y / k) * Wo + xx / k) * C + c) * inv;
        io<T>::st(dx + i, v);
    }
}

// ---------------------------------------------------------------- BatchNorm2d, training mode
// Column statistics of x [R, C]: every block sums a fixed slab of rows (shifted by the column's first value, so that
// E[x^2] - E[x]^2 does not cancel), partial[block][2][C]; bn_finish combines them in block order.
template <typename T>
__global__ __launch_bounds__(256) void bn_partial_kernel(const T* x, int64_t R, int C, int64_t rows_per_block, float* partial) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
    const float k = io<T>::ld(x + c);
    float s1 = 0.f, s2 = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
        const float d = io<T>::ld(x + r * C + c) - k;
        s1 += d;
        s2 += d * d;
    }
    partial[((int64_t)blockIdx.y * 2 + 0) * C + c] = s1;
    partial[((int64_t)blockIdx.y * 2 + 1) * C + c] = s2;
}
// stats_out[0..C) = sum(x - k), [C..2C) = sum (x - k)^2, [2C..3C) = k   (what ranks exchange for a synchronised BatchNorm)
template <typename T>
__global__ __launch_bounds__(256) void bn_collect_kernel(const T* x, const float* partial, int nblocks, int C, float* stats_out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s1 = 0.f, s2 = 0.f;
    for (int b = 0; b < nblocks; ++b) {
        s1 += partial[((int64_t)b * 2 + 0) * C + c];
        s2 += partial[((int64_t)b * 2 + 1) * C + c];
    }
    stats_out[c] = s1;
    stats_out[C + c] = s2;
    stats_out[2 * C + c] = io<T>::ld(x + c);
}
// From `nparts` statistic triples (one per rank, count rows each): batch mean and 1/sqrt(biased var + eps); running statistics
// as torch (momentum update with the UNBIASED variance).
__global__ __launch_bounds__(256) void bn_finish_kernel(const float* stats, int nparts, int C, float count, float eps, float momentum, float* mean,
                                                        float* rstd, float* running_mean, float* running_var) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    // combine shifted sums: mean_p = k_p + s1_p / n; M2_p = s2_p - s1_p^2 / n; Chan's pairwise update in part order
    float n_tot = 0.f, mu = 0.f, m2 = 0.f;
    for (int p = 0; p < nparts; ++p) {
        const float* s = stats + (int64_t)p * 3 * C;
        const float mp = s[2 * C + c] + s[c] / count, m2p = s[C + c] - s[c] * s[c] / count;
        const float d = mp - mu, n_new = n_tot + count;
        mu += d * (count / n_new);
        m2 += m2p + d * d * (n_tot * count / n_new);
        n_tot = n_new;
    }
    const float var = fmaxf(m2 / n_tot, 0.f);
    mean[c] = mu;
    rstd[c] = 1.0f / sqrtf(var + eps);
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (m2 / fmaxf(n_tot - 1.f, 1.f));
    }
}
// y = act((x - mean) * rstd * gamma + beta (+ res)),  act = ReLU or identity
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                                       const T* res, int64_t n, int C, int relu, T* y) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        float v = (io<T>::ld(x + i) - mean[c]) * rstd[c] * gamma[c] + beta[c];
        if (res) v += io<T>::ld(res + i);
        if (relu) v = fmaxf(v, 0.f);
        io<T>::st(y + i, v);
    }
}
// g = dy masked by the ReLU (y > 0); partial[block][2][C] = sum g, sum g * xhat over the block's rows
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const T* dy, const T* y, const T* x, const float* mean, const float* rstd, int64_t R, int C,
                                                             int64_t rows_per_block, int relu, float* partial) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
    const float mu = mean[c], rs = rstd[c];
    float sg = 0.f, sgx = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
        float g = io<T>::ld(dy + r * C + c);
        if (relu && !(io<T>::ld(y + r * C + c) > 0.f)) g = 0.f;
        sg += g;
        sgx += g * (io<T>::ld(x + r * C + c) - mu) * rs;
    }
    partial[((int64_t)blockIdx.y * 2 + 0) * C + c] = sg;
    partial[((int64_t)blockIdx.y * 2 + 1) * C + c] = sgx;
}
__global__ __launch_bounds__(256) void bn_bwd_collect_kernel(const float* partial, int nblocks, int C, float* sums /* [2C]: sum g | sum g xhat */) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float a = 0.f, b = 0.f;
    for (int k = 0; k < nblocks; ++k) {
        a += partial[((int64_t)k * 2 + 0) * C + c];
        b += partial[((int64_t)k * 2 + 1) * C + c];
    }
    sums[c] = a;
    sums[C + c] = b;
}
// dx = gamma * rstd * (g - sum_g / N - xhat * sum_gx / N);  dres (optional) = g;  dgamma (+)= sum_gx, dbeta (+)= sum_g  (block 0 writes them)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* dy, const T* y, const T* x, const float* mean, const float* rstd, const float* gamma,
                                                           const float* sums, float count, int64_t n, int C, int relu, int accumulate, T* dx, T* dres,
                                                           float* dgamma, float* dbeta) {
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < C; c += 256) {
            dgamma[c] = (accumulate ? dgamma[c] : 0.f) + sums[C + c];
            dbeta[c] = (accumulate ? dbeta[c] : 0.f) + sums[c];
        }
    const float inv = 1.0f / count;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        float g = io<T>::ld(dy + i);
        if (relu && !(io<T>::ld(y + i) > 0.f)) g = 0.f;
        const float xh = (io<T>::ld(x + i) - mean[c]) * rstd[c];
        io<T>::st(dx + i, gamma[c] * rstd[c] * (g - sums[c] * inv - xh * sums[C + c] * inv));
        if (dres) io<T>::st(dres + i, g);
    }
}

// ---------------------------------------------------------------- attention pool: tokens = [mean over positions; positions] + positional embedding
template <typename T>
__global__ __launch_bounds__(256) void attnpool_tokens_fwd_kernel(const T* x, const float* pos, int B, int HW, int C, T* t) {
    const int64_t total = (int64_t)B * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t b = i / C;
        float s = 0.f;
        for (int p = 0; p < HW; ++p) {
            const float v = io<T>::ld(x + ((int64_t)b * HW + p) * C + c);
            s += v;
            io<T>::st(t + ((int64_t)b * (HW + 1) + p + 1) * C + c, v + pos[(int64_t)(p + 1) * C + c]);
        }
        io<T>::st(t + (int64_t)b * (HW + 1) * C + c, s / (float)HW + pos[c]);
    }
}
// dx[b][p] = dt[b][p+1] + dt[b][0] / HW
template <typename T>
__global__ __launch_bounds__(256) void attnpool_tokens_bwd_kernel(const T* dt, int B, int HW, int C, T* dx) {
    const int64_t total = (int64_t)B * HW * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t r = i / C;
        const int64_t b = r / HW, p = r % HW;
        io<T>::st(dx + i, io<T>::ld(dt + ((int64_t)b * (HW + 1) + p + 1) * C + c) + io<T>::ld(dt + (int64_t)b * (HW + 1) * C + c) / (float)HW);
    }
}


// ================================================================ 4-channel-per-thread forms (C % 4 == 0): 8- / 16-byte accesses
// The scalar kernels above are the general forms (any C; the 3-channel image input); the step runs these.
template <typename T>
__global__ __launch_bounds__(256) void im2col3x3_v4_kernel(const T* x, int B, int H, int W, int C, int stride, int Ho, int Wo, int kpad, T* out) {
    const int k4 = kpad >> 2;
    const int64_t total = (int64_t)B * Ho * Wo * k4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int col = (int)(i % k4) * 4;
        const int64_t row = i / k4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (col < 9 * C) {
            const int tap = col / C, c = col - tap * C;
            const int xo = (int)(row % Wo), yo = (int)((row / Wo) % Ho), b = (int)(row / ((int64_t)Wo * Ho));
            const int y = yo * stride + tap / 3 - 1, xx = xo * stride + tap % 3 - 1;
            if (y >= 0 && y < H && xx >= 0 && xx < W) v = io<T>::ld4(x + (((int64_t)b * H + y) * W + xx) * C + c);
        }
        io<T>::st4(out + row * kpad + col, v);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void col2im3x3_v4_kernel(const T* dcols, int B, int H, int W, int C, int stride, int Ho, int Wo, int kpad, T* dx) {
    const int c4n = C >> 2;
    const int64_t total = (int64_t)B * H * W * c4n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % c4n) * 4;
        const int64_t p = i / c4n;
        const int xx = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((int64_t)W * H));
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = y + 1 - ky;
            if (ty < 0 || ty % stride != 0 || ty / stride >= Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = xx + 1 - kx;
                if (tx < 0 || tx % stride != 0 || tx / stride >= Wo) continue;
                s += io<T>::ld4(dcols + (((int64_t)b * Ho + ty / stride) * Wo + tx / stride) * kpad + (3 * ky + kx) * C + c);
            }
        }
        io<T>::st4(dx + p * C + c, s);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_v4_kernel(const T* x, int B, int H, int W, int C, int k, T* y) {
    const int Ho = H / k, Wo = W / k, c4n = C >> 2;
    const int64_t total = (int64_t)B * Ho * Wo * c4n;
    const float inv = 1.0f / (float)(k * k);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % c4n) * 4;
        const int64_t p = i / c4n;
        const int xo = (int)(p % Wo), yo = (int)((p / Wo) % Ho), b = (int)(p / ((int64_t)Wo * Ho));
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx) s += io<T>::ld4(x + (((int64_t)b * H + yo * k + dy) * W + xo * k + dx) * C + c);
        io<T>::st4(y + p * C + c, s * inv);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_v4_kernel(const T* dy, int B, int H, int W, int C, int k, T* dx) {
    const int Ho = H / k, Wo = W / k, c4n = C >> 2;
    const int64_t total = (int64_t)B * H * W * c4n;
    const float inv = 1.0f / (float)(k * k);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % c4n) * 4;
        const int64_t p = i / c4n;
        const int xx = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((int64_t)W * H));
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (y / k < Ho && xx / k < Wo) v = io<T>::ld4(dy + (((int64_t)b * Ho + y / k) * Wo + xx / k) * C + c) * inv;
        io<T>::st4(dx + p * C + c, v);
    }
}

// Column reductions over x [R, C]: the 256 threads of a block form RPI = 256 / LPR row lanes of LPR = chunk / 4 four-channel lanes
// (chunk = min(C, 1024) channels per blockIdx.x); every thread accumulates its four channels over every RPI-th row of the block's
// row slab, the row lanes are then summed through LDS in lane order.  kind 0: sum (x - k), sum (x - k)^2 with k = x[0][c];
// kind 1 (backward): sum g, sum g * xhat with g = dy masked by y > 0.
template <typename T, int KIND>
__global__ __launch_bounds__(256) void bn_partial_v4_kernel(const T* x, const T* dy, const T* y, const float* mean, const float* rstd, const float* gamma,
                                                            const float* beta, int64_t R, int C, int lpr, int64_t rows_per_block, int relu, float* partial) {
    __shared__ f32x4 sm[2][256];
    const int t = threadIdx.x, lc = t % lpr, lr = t / lpr, rpi = 256 / lpr;
    const int c = blockIdx.x * (lpr * 4) + lc * 4;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    f32x4 k = {0.f, 0.f, 0.f, 0.f}, rs = {0.f, 0.f, 0.f, 0.f};
    f32x4 gm = {0.f, 0.f, 0.f, 0.f}, bt = {0.f, 0.f, 0.f, 0.f};
    if (KIND == 0) k = io<T>::ld4(x + c);
    else {
        k = *(const f32x4*)(mean + c); rs = *(const f32x4*)(rstd + c);
        if (relu && !y) { gm = *(const f32x4*)(gamma + c); bt = *(const f32x4*)(beta + c); }
    }
    auto add = [&](const f32x4& xv, f32x4 g, const f32x4& yv) {
        if (KIND == 0) {
            const f32x4 d = xv - k;
            a += d;
            b += d * d;
        } else {
            const f32x4 xh = (xv - k) * rs;
            if (relu) {   // the ReLU's mask: from the stored output, or (no residual in the forward) recomputed from x - one tensor less to read
                const f32x4 m = y ? yv : xh * gm + bt;
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = m[j] > 0.f ? g[j] : 0.f;
            }
            a += g;
            b += g * xh;
        }
    };
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    int64_t r = r0 + lr;
    for (; r + 3 * rpi < r1; r += 4 * rpi) {   // four rows in flight per lane: the loads first, then the (ordered) accumulation
        f32x4 xv[4], gv[4], yv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t e = (r + u * rpi) * C + c;
            xv[u] = io<T>::ld4(x + e);
            gv[u] = KIND == 1 ? io<T>::ld4(dy + e) : zero;
            yv[u] = KIND == 1 && relu && y ? io<T>::ld4(y + e) : zero;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) add(xv[u], gv[u], yv[u]);
    }
    for (; r < r1; r += rpi) {
        const int64_t e = r * C + c;
        add(io<T>::ld4(x + e), KIND == 1 ? io<T>::ld4(dy + e) : zero, KIND == 1 && relu && y ? io<T>::ld4(y + e) : zero);
    }
    sm[0][t] = a; sm[1][t] = b;
    __syncthreads();
    if (lr == 0) {
        for (int j = 1; j < rpi; ++j) { a += sm[0][j * lpr + lc]; b += sm[1][j * lpr + lc]; }
        *(f32x4*)(partial + ((int64_t)blockIdx.y * 2 + 0) * C + c) = a;
        *(f32x4*)(partial + ((int64_t)blockIdx.y * 2 + 1) * C + c) = b;
    }
}
// out[0..C) = sum_b partial[b][0], out[C..2C) = sum_b partial[b][1]: one block per four channels, thread t sums the row blocks t, t + 256, ...
// (at most four loads each), then a fixed-order tree through LDS; shift != null: out[2C..3C) = shift values x[0][c] (forward statistics)
template <typename T>
__global__ __launch_bounds__(256) void bn_collect_v_kernel(const float* partial, int nblocks, int C, const T* shift_row, float* out) {
    __shared__ f32x4 sm[2][256];
    const int t = threadIdx.x, c = blockIdx.x * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    for (int k = t; k < nblocks; k += 256) {
        a += *(const f32x4*)(partial + ((int64_t)k * 2 + 0) * C + c);
        b += *(const f32x4*)(partial + ((int64_t)k * 2 + 1) * C + c);
    }
    sm[0][t] = a; sm[1][t] = b;
    __syncthreads();
#pragma unroll
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) { sm[0][t] += sm[0][t + o]; sm[1][t] += sm[1][t + o]; }
        __syncthreads();
    }
    if (t == 0) {
        *(f32x4*)(out + c) = sm[0][0];
        *(f32x4*)(out + C + c) = sm[1][0];
        if (shift_row) {
#pragma unroll
            for (int j = 0; j < 4; ++j) out[2 * C + c + j] = io<T>::ld(shift_row + c + j);
        }
    }
}
// halo_w != 0: the output goes to a zero-bordered NHWC image [B][halo_h + 2][halo_w + 2][CS] (row r = (b, y, x) -> pixel (y + 1, x + 1)),
// the layout sc_conv3x3_bf16 reads; CS >= C is the image's channel count (a 32-channel activation inside a 64-channel image, the
// implicit GEMM's K granularity); the border and the channels >= C are zeroed by the caller
__device__ __forceinline__ int64_t halo_elem(int64_t e, int C, int hh, int hw, int CS) {
    if (!hw) return e;
    const int64_t r = e / C;
    const int c = (int)(e - r * C);
    const int xx = (int)(r % hw);
    const int64_t t = r / hw;
    const int yy = (int)(t % hh);
    const int64_t b = t / hh;
    return (((b * (hh + 2) + yy + 1) * (hw + 2)) + xx + 1) * CS + c;
}
// INV: the grid stride is a multiple of C, so a thread stays on its four channels for the whole loop and the per-channel vectors are
// loaded once (the launchers check it); otherwise they are re-read (L2 / cache hits) every iteration.
template <typename T, bool INV>
__global__ __launch_bounds__(256) void bn_apply_v4_kernel(const T* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                                          const T* res, int64_t n4, int C, int relu, int hh, int hw, int hc, T* y) {
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, step = (int64_t)gridDim.x * 256;
    f32x4 mu, rs, g, bt;      // the expression below is the one the backward kernels re-evaluate for the ReLU mask: keep them identical
    auto params = [&](int c) {
        mu = *(const f32x4*)(mean + c); rs = *(const f32x4*)(rstd + c);
        g = *(const f32x4*)(gamma + c); bt = *(const f32x4*)(beta + c);
    };
    if (INV) params((int)((i0 * 4) % C));
    auto one = [&](int64_t i, f32x4 xv, f32x4 rv) {
        f32x4 v = (xv - mu) * rs * g + bt;
        if (res) v += rv;
        if (relu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        io<T>::st4(y + halo_elem(i * 4, C, hh, hw, hc), v);
    };
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    int64_t i = i0;
    if (INV)
        for (; i + step < n4; i += 2 * step) {      // two quads in flight per lane
            const f32x4 x0 = io<T>::ld4(x + i * 4), x1 = io<T>::ld4(x + (i + step) * 4);
            const f32x4 r0 = res ? io<T>::ld4(res + i * 4) : zero, r1 = res ? io<T>::ld4(res + (i + step) * 4) : zero;
            one(i, x0, r0);
            one(i + step, x1, r1);
        }
    for (; i < n4; i += step) {
        if (!INV) params((int)((i * 4) % C));
        one(i, io<T>::ld4(x + i * 4), res ? io<T>::ld4(res + i * 4) : zero);
    }
}
template <typename T, bool INV>
__global__ __launch_bounds__(256) void bn_bwd_apply_v4_kernel(const T* dy, const T* y, const T* x, const float* mean, const float* rstd, const float* gamma,
                                                              const float* beta, const float* sums, float count, int64_t n4, int C, int relu, int accumulate,
                                                              int hh, int hw, int hc, T* dx, T* dres, float* dgamma, float* dbeta) {
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < C; c += 256) {
            dgamma[c] = (accumulate ? dgamma[c] : 0.f) + sums[C + c];
            dbeta[c] = (accumulate ? dbeta[c] : 0.f) + sums[c];
        }
    const float inv = 1.0f / count;
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, step = (int64_t)gridDim.x * 256;
    f32x4 mu, rs, gm, bt, sg, sgx;
    auto params = [&](int c) {
        mu = *(const f32x4*)(mean + c); rs = *(const f32x4*)(rstd + c); gm = *(const f32x4*)(gamma + c);
        sg = *(const f32x4*)(sums + c); sgx = *(const f32x4*)(sums + C + c);
        bt = relu && !y ? *(const f32x4*)(beta + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    if (INV) params((int)((i0 * 4) % C));
    auto one = [&](int64_t i, f32x4 g, f32x4 xv, f32x4 yv) {
        const f32x4 xh = (xv - mu) * rs;
        if (relu) {
            const f32x4 m = y ? yv : xh * gm + bt;
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = m[j] > 0.f ? g[j] : 0.f;
        }
        io<T>::st4(dx + halo_elem(i * 4, C, hh, hw, hc), gm * rs * (g - sg * inv - xh * sgx * inv));
        if (dres) io<T>::st4(dres + i * 4, g);
    };
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const bool rd_y = relu && y;
    int64_t i = i0;
    if (INV)
        for (; i + step < n4; i += 2 * step) {
            const f32x4 g0 = io<T>::ld4(dy + i * 4), g1 = io<T>::ld4(dy + (i + step) * 4);
            const f32x4 x0 = io<T>::ld4(x + i * 4), x1 = io<T>::ld4(x + (i + step) * 4);
            const f32x4 y0 = rd_y ? io<T>::ld4(y + i * 4) : zero, y1 = rd_y ? io<T>::ld4(y + (i + step) * 4) : zero;
            one(i, g0, x0, y0);
            one(i + step, g1, x1, y1);
        }
    for (; i < n4; i += step) {
        if (!INV) params((int)((i * 4) % C));
        one(i, io<T>::ld4(dy + i * 4), io<T>::ld4(x + i * 4), rd_y ? io<T>::ld4(y + i * 4) : zero);
    }
}

// four-channel lanes per row for the column reductions, or 0 when the channel count does not fit the vector form
int bn_lpr(int64_t c) {
    if (c % 4 != 0) return 0;
    const int64_t chunk = c <= 1024 ? c : 1024;
    if (c % chunk != 0) return 0;
    const int lpr = (int)(chunk / 4);
    return (lpr & (lpr - 1)) == 0 ? lpr : 0;
}

// SC_CONV_SCALAR=<bit mask>: force the general (scalar) kernels - 1 im2col, 2 col2im, 4 average pools, 8 BatchNorm statistics, 16 BatchNorm
// apply, 32 BatchNorm backward sums, 64 BatchNorm backward apply (A/B and bisection knob)
bool vec_ok(int bit) {
    static const int mask = [] { const char* e = sc_debug_env("SC_CONV_SCALAR"); return e ? atoi(e) : 0; }();
    return !(mask & bit);
}
unsigned stream_grid(int64_t total) { return (unsigned)min((int64_t)4096, max((int64_t)1, sc_cdiv(total, 256))); }
int bn_blocks(int64_t rows) { return (int)min((int64_t)BN_MAX_BLOCKS, max((int64_t)1, sc_cdiv(rows, 64))); }

}  // namespace

#define SC_DT(dtype, CALL_BF16, CALL_F32)                                                        \
    do {                                                                                         \
        if ((dtype) == SC_BF16) { CALL_BF16; }                                                   \
        else if ((dtype) == SC_F32) { CALL_F32; }                                                \
        else return sc_set_error(SC_ERR_DTYPE, "conv: bad dtype %d", (int)(dtype));               \
    } while (0)

extern "C" int sc_im2col3x3(const void* x, int in_nchw_f32, int dtype, int64_t batch, int64_t h, int64_t w, int64_t c, int64_t stride, int64_t kpad,
                            void* out, void* stream) {
    SC_REQUIRE(x && out && batch > 0 && h > 0 && w > 0 && c > 0 && (stride == 1 || stride == 2) && kpad >= 9 * c, SC_ERR_ARG, "sc_im2col3x3: bad argument");
    const int Ho = (int)((h - 1) / stride + 1), Wo = (int)((w - 1) / stride + 1);
    const int64_t total = batch * Ho * Wo * kpad;
    hipStream_t st = (hipStream_t)stream;
#define IM(TI, TO, N) hipLaunchKernelGGL((im2col3x3_kernel<TI, TO, N>), dim3(stream_grid(total)), dim3(256), 0, st, (const TI*)x, (int)batch, (int)h, (int)w, (int)c, (int)stride, Ho, Wo, (int)kpad, (TO*)out)
#define IMV(T) hipLaunchKernelGGL(im2col3x3_v4_kernel<T>, dim3(stream_grid(total / 4)), dim3(256), 0, st, (const T*)x, (int)batch, (int)h, (int)w, (int)c, (int)stride, Ho, Wo, (int)kpad, (T*)out)
    if (in_nchw_f32) SC_DT(dtype, IM(float, bf16_t, true), IM(float, float, true));
    else if (c % 4 == 0 && kpad % 4 == 0 && vec_ok(1)) SC_DT(dtype, IMV(bf16_t), IMV(float));
    else SC_DT(dtype, IM(bf16_t, bf16_t, false), IM(float, float, false));
#undef IMV
#undef IM
    SC_CHECK_LAUNCH();
    return SC_OK;
}
extern "C" int sc_col2im3x3(const void* dcols, int dtype, int64_t batch, int64_t h, int64_t w, int64_t c, int64_t stride, int64_t kpad, void* dx,
                            void* stream) {
    SC_REQUIRE(dcols && dx && batch > 0 && h > 0 && w > 0 && c > 0 && (stride == 1 || stride == 2) && kpad >= 9 * c, SC_ERR_ARG, "sc_col2im3x3: bad argument");
    const int Ho = (int)((h - 1) / stride + 1), Wo = (int)((w - 1) / stride + 1);
    const int64_t total = batch * h * w * c;
    hipStream_t st = (hipStream_t)stream;
#define CI(T) hipLaunchKernelGGL(col2im3x3_kernel<T>, dim3(stream_grid(total)), dim3(256), 0, st, (const T*)dcols, (int)batch, (int)h, (int)w, (int)c, (int)stride, Ho, Wo, (int)kpad, (T*)dx)
#define CIV(T) hipLaunchKernelGGL(col2im3x3_v4_kernel<T>, dim3(stream_grid(total / 4)), dim3(256), 0, st, (const T*)dcols, (int)batch, (int)h, (int)w, (int)c, (int)stride, Ho, Wo, (int)kpad, (T*)dx)
    if (c % 4 == 0 && kpad % 4 == 0 && vec_ok(2)) SC_DT(dtype, CIV(bf16_t), CIV(float));
    else SC_DT(dtype, CI(bf16_t), CI(float));
#undef CIV
#undef CI
    SC_CHECK_LAUNCH();
    return SC_OK;
}
extern "C" int sc_avgpool_fwd(const void* x, int dtype, int64_t batch, int64_t h, int64_t w, int64_t c, int64_t k, void* y, void* stream) {
    SC_REQUIRE(x && y && batch > 0 && k >= 1 && h % k == 0 && w % k == 0 && c > 0, SC_ERR_ARG, "sc_avgpool_fwd: bad argument");
    const int64_t total = batch * (h / k) * (w / k) * c;
    hipStream_t st = (hipStream_t)stream;
#define AP(T) hipLaunchKernelGGL(avgpool_fwd_kernel<T>, dim3(stream_grid(total)), dim3(256), 0, st, (const T*)x, (int)batch, (int)h, (int)w, (int)c, (int)k, (T*)y)
#define APV(T) hipLaunchKernelGGL(avgpool_fwd_v4_kernel<T>, dim3(stream_grid(total / 4)), dim3(256), 0, st, (const T*)x, (int)batch, (int)h, (int)w, (int)c, (int)k, (T*)y)
    if (c % 4 == 0 && vec_ok(4)) SC_DT(dtype, APV(bf16_t), APV(float));
    else SC_DT(dtype, AP(bf16_t), AP(float));
#undef APV
#undef AP
    SC_CHECK_LAUNCH();
    return SC_OK;
}
extern "C" int sc_avgpool_bwd(const void* dy, int dtype, int64_t batch, int64_t h, int64_t w, int64_t c, int64_t k, void* dx, void* stream) {
    SC_REQUIRE(dy && dx && batch > 0 && k >= 1 && h % k == 0 && w % k == 0 && c > 0, SC_ERR_ARG, "sc_avgpool_bwd: bad argument");
    const int64_t total = batch * h * w * c;
    hipStream_t st = (hipStream_t)stream;
#define AP(T) hipLaunchKernelGGL(avgpool_bwd_kernel<T>, dim3(stream_grid(total)), dim3(256), 0, st, (const T*)dy, (int)batch, (int)h, (int)w, (int)c, (int)k, (T*)dx)
#define APV(T) hipLaunchKernelGGL(avgpool_bwd_v4_kernel<T>, dim3(stream_grid(total / 4)), dim3(256), 0, st, (const T*)dy, (int)batch, (int)h, (int)w, (int)c, (int)k, (T*)dx)
    if (c % 4 == 0 && vec_ok(4)) SC_DT(dtype, APV(bf16_t), APV(float));
    else SC_DT(dtype, AP(bf16_t), AP(float));
#undef APV
#undef AP
    SC_CHECK_LAUNCH();
    return SC_OK;
}

extern "C" size_t sc_bn_workspace_bytes(int64_t rows, int64_t c) { return (size_t)bn_blocks(rows) * 2 * (size_t)c * sizeof(float); }

extern "C" int sc_bn_stats(const void* x, int dtype, int64_t rows, int64_t c, float* stats, void* ws, size_t ws_bytes, void* stream) {
    SC_REQUIRE(x && stats && ws && rows > 0 && c > 0, SC_ERR_ARG, "sc_bn_stats: bad argument");
    SC_REQUIRE(ws_bytes >= sc_bn_workspace_bytes(rows, c), SC_ERR_WORKSPACE, "sc_bn_stats: workspace too small");
    const int nb = bn_blocks(rows);
    const int64_t rpb = sc_cdiv(rows, nb);
    const int nbe = (int)sc_cdiv(rows, rpb);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)sc_cdiv(c, 256), (unsigned)nbe);
#define BS(T)                                                                                                                              \
    do {                                                                                                                                   \
        hipLaunchKernelGGL(bn_partial_kernel<T>, grid, dim3(256), 0, st, (const T*)x, rows, (int)c, rpb, (float*)ws);                      \
        hipLaunchKernelGGL(bn_collect_kernel<T>, dim3((unsigned)sc_cdiv(c, 256)), dim3(256), 0, st, (const T*)x, (const float*)ws, nbe, (int)c, stats); \
    } while (0)
#define BSV(T)                                                                                                                             \
    do {                                                                                                                                   \
        hipLaunchKernelGGL((bn_partial_v4_kernel<T, 0>), dim3((unsigned)(c / (lpr * 4)), (unsigned)nbe), dim3(256), 0, st, (const T*)x, (const T*)nullptr, \
                           (const T*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, rows, (int)c, lpr, \
                           rpb, 0, (float*)ws);                                                                                            \
        hipLaunchKernelGGL(bn_collect_v_kernel<T>, dim3((unsigned)(c / 4)), dim3(256), 0, st, (const float*)ws, nbe, (int)c, (const T*)x, stats); \
    } while (0)
    const int lpr = vec_ok(8) ? bn_lpr(c) : 0;
    if (lpr) SC_DT(dtype, BSV(bf16_t), BSV(float));
    else SC_DT(dtype, BS(bf16_t), BS(float));
#undef BSV
#undef BS
    SC_CHECK_LAUNCH();
    return SC_OK;
}
extern "C" int sc_bn_finish(const float* stats, int64_t nparts, int64_t c, int64_t rows_per_part, float eps, float momentum, float* mean, float* rstd,
                            float* running_mean, float* running_var, void* stream) {
    SC_REQUIRE(stats && mean && rstd && nparts > 0 && c > 0 && rows_per_part > 0 && (running_mean == nullptr) == (running_var == nullptr), SC_ERR_ARG,
               "sc_bn_finish: bad argument");
    hipLaunchKernelGGL(bn_finish_kernel, dim3((unsigned)sc_cdiv(c, 256)), dim3(256), 0, (hipStream_t)stream, stats, (int)nparts, (int)c, (float)rows_per_part,
                       eps, momentum, mean, rstd, running_mean, running_var);
    SC_CHECK_LAUNCH();
    return SC_OK;
}
extern "C" int sc_bn_apply(const void* x, int dtype, int64_t rows, int64_t c, const float* mean, const float* rstd, const float* gamma, const float* beta,
                           const void* res, int relu, int64_t halo_h, int64_t halo_w, int64_t halo_c, void* y, void* stream) {
    SC_REQUIRE(x && y && mean && rstd && gamma && beta && rows > 0 && c > 0, SC_ERR_ARG, "sc_bn_apply: bad argument");
    SC_REQUIRE((halo_h == 0) == (halo_w == 0) && (halo_w == 0 || (c % 4 == 0 && vec_ok(16) && rows % (halo_h * halo_w) == 0 && halo_c >= c && halo_c % 4 == 0)),
               SC_ERR_SHAPE, "sc_bn_apply: the bordered output needs c %% 4 == 0, rows = batch * halo_h * halo_w and halo_c >= c");
    const int64_t n = rows * c;
    hipStream_t st = (hipStream_t)stream;
#define BA(T) hipLaunchKernelGGL(bn_apply_kernel<T>, dim3(stream_grid(n)), dim3(256), 0, st, (const T*)x, mean, rstd, gamma, beta, (const T*)res, n, (int)c, relu, (T*)y)
#define BAV_(T, INV) hipLaunchKernelGGL((bn_apply_v4_kernel<T, INV>), dim3(stream_grid(n / 4)), dim3(256), 0, st, (const T*)x, mean, rstd, gamma, beta, (const T*)res, n / 4, (int)c, relu, (int)halo_h, (int)halo_w, (int)halo_c, (T*)y)
#define BAV(T) do { if (((int64_t)stream_grid(n / 4) * 1024) % c == 0) BAV_(T, true); else BAV_(T, false); } while (0)
    if (c % 4 == 0 && vec_ok(16)) SC_DT(dtype, BAV(bf16_t), BAV(float));
    else SC_DT(dtype, BA(bf16_t), BA(float));
#undef BAV
#undef BAV_
#undef BA
    SC_CHECK_LAUNCH();
    return SC_OK;
}
extern "C" int sc_bn_bwd_stats(const void* dy, const void* y, const void* x, int dtype, int64_t rows, int64_t c, const float* mean, const float* rstd,
                               const float* gamma, const float* beta, int relu, float* sums, void* ws, size_t ws_bytes, void* stream) {
    SC_REQUIRE(dy && x && sums && ws && mean && rstd && rows > 0 && c > 0, SC_ERR_ARG, "sc_bn_bwd_stats: bad argument");
    SC_REQUIRE(!relu || y || (gamma && beta && bn_lpr(c) && vec_ok(32)), SC_ERR_ARG,
               "sc_bn_bwd_stats: ReLU needs the stored output y (or gamma and beta, for a channel count the four-channel kernels take)");
    SC_REQUIRE(ws_bytes >= sc_bn_workspace_bytes(rows, c), SC_ERR_WORKSPACE, "sc_bn_bwd_stats: workspace too small");
    const int nb = bn_blocks(rows);
    const int64_t rpb = sc_cdiv(rows, nb);
    const int nbe = (int)sc_cdiv(rows, rpb);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)sc_cdiv(c, 256), (unsigned)nbe);
#define BB(T) hipLaunchKernelGGL(bn_bwd_partial_kernel<T>, grid, dim3(256), 0, st, (const T*)dy, (const T*)y, (const T*)x, mean, rstd, rows, (int)c, rpb, relu, (float*)ws)
#define BBV(T) hipLaunchKernelGGL((bn_partial_v4_kernel<T, 1>), dim3((unsigned)(c / (lpr * 4)), (unsigned)nbe), dim3(256), 0, st, (const T*)x, (const T*)dy, (const T*)y, mean, rstd, gamma, beta, rows, (int)c, lpr, rpb, relu, (float*)ws)
    const int lpr = vec_ok(32) ? bn_lpr(c) : 0;
    if (lpr) {
        SC_DT(dtype, BBV(bf16_t), BBV(float));
        hipLaunchKernelGGL(bn_collect_v_kernel<float>, dim3((unsigned)(c / 4)), dim3(256), 0, st, (const float*)ws, nbe, (int)c, (const float*)nullptr, sums);
    } else {
        SC_DT(dtype, BB(bf16_t), BB(float));
        hipLaunchKernelGGL(bn_bwd_collect_kernel, dim3((unsigned)sc_cdiv(c, 256)), dim3(256), 0, st, (const float*)ws, nbe, (int)c, sums);
    }
#undef BBV
#undef BB
    SC_CHECK_LAUNCH();
    return SC_OK;
}
extern "C" int sc_bn_bwd_apply(const void* dy, const void* y, const void* x, int dtype, int64_t rows, int64_t c, const float* mean, const float* rstd,
                               const float* gamma, const float* beta, const float* sums, int64_t total_rows, int relu, int accumulate,
                               int64_t halo_h, int64_t halo_w, int64_t halo_c, void* dx, void* dres, float* dgamma, float* dbeta, void* stream) {
    SC_REQUIRE(dy && x && dx && sums && mean && rstd && gamma && dgamma && dbeta && rows > 0 && c > 0 && total_rows >= rows, SC_ERR_ARG,
               "sc_bn_bwd_apply: bad argument");
    SC_REQUIRE((halo_h == 0) == (halo_w == 0) && (halo_w == 0 || (c % 4 == 0 && vec_ok(64) && rows % (halo_h * halo_w) == 0 && halo_c >= c && halo_c % 4 == 0)),
               SC_ERR_SHAPE, "sc_bn_bwd_apply: the bordered dx needs c %% 4 == 0, rows = batch * halo_h * halo_w and halo_c >= c");
    SC_REQUIRE(!relu || y || (beta && c % 4 == 0 && vec_ok(64)), SC_ERR_ARG, "sc_bn_bwd_apply: ReLU needs the stored output y (or beta, with c % 4 == 0)");
    const int64_t n = rows * c;
    hipStream_t st = (hipStream_t)stream;
#define BA(T) hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(stream_grid(n)), dim3(256), 0, st, (const T*)dy, (const T*)y, (const T*)x, mean, rstd, gamma, sums, (float)total_rows, n, (int)c, relu, accumulate, (T*)dx, (T*)dres, dgamma, dbeta)
#define BAV_(T, INV) hipLaunchKernelGGL((bn_bwd_apply_v4_kernel<T, INV>), dim3(stream_grid(n / 4)), dim3(256), 0, st, (const T*)dy, (const T*)y, (const T*)x, mean, rstd, gamma, beta, sums, (float)total_rows, n / 4, (int)c, relu, accumulate, (int)halo_h, (int)halo_w, (int)halo_c, (T*)dx, (T*)dres, dgamma, dbeta)
#define BAV(T) do { if (((int64_t)stream_grid(n / 4) * 1024) % c == 0) BAV_(T, true); else BAV_(T, false); } while (0)
    if (c % 4 == 0 && vec_ok(64)) SC_DT(dtype, BAV(bf16_t), BAV(float));
    else SC_DT(dtype, BA(bf16_t), BA(float));
#undef BAV
#undef BAV_
#undef BA
    SC_CHECK_LAUNCH();
    return SC_OK;
}
extern "C" int sc_attnpool_tokens_fwd(const void* x, int dtype, const float* pos, int64_t batch, int64_t hw, int64_t c, void* tokens, void* stream) {
    SC_REQUIRE(x && pos && tokens && batch > 0 && hw > 0 && c > 0, SC_ERR_ARG, "sc_attnpool_tokens_fwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
#define AT(T) hipLaunchKernelGGL(attnpool_tokens_fwd_kernel<T>, dim3(stream_grid(batch * c)), dim3(256), 0, st, (const T*)x, pos, (int)batch, (int)hw, (int)c, (T*)tokens)
    SC_DT(dtype, AT(bf16_t), AT(float));
#undef AT
    SC_CHECK_LAUNCH();
    return SC_OK;
}
extern "C" int sc_attnpool_tokens_bwd(const void* dtokens, int dtype, int64_t batch, int64_t hw, int64_t c, void* dx, void* stream) {
    SC_REQUIRE(dtokens && dx && batch > 0 && hw > 0 && c > 0, SC_ERR_ARG, "sc_attnpool_tokens_bwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
#define AT(T) hipLaunchKernelGGL(attnpool_tokens_bwd_kernel<T>, dim3(stream_grid(batch * hw * c)), dim3(256), 0, st, (const T*)dtokens, (int)batch, (int)hw, (int)c, (T*)dx)
    SC_DT(dtype, AT(bf16_t), AT(float));
#undef AT
    SC_CHECK_LAUNCH();
    return SC_OK;
}
