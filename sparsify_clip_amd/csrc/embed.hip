// Stem / head data movement of the two towers: im2col for the patch-embedding GEMM, cls/positional token assembly,
// token-embedding gather and its deterministic scatter-add, pooled-row gather/scatter, dtype casts and the
// [in,out] bf16 weight copies.  All HBM-bound, coalesced along the contiguous dimension.
#include "common.h"

namespace {

// patches[(b,py,px)][(c,ky,kx)] = img[b][c][py*P+ky][px*P+kx]; columns >= 3*P*P are zero padding
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const float* img, int res, int patch, int g, int kreal, int kpad, T* out) {
    const int64_t row = blockIdx.x;   // b*g*g + py*g + px
    const int b = (int)(row / (g * g)), pp = (int)(row % (g * g)), py = pp / g, px = pp % g;
    const float* src = img + (int64_t)b * 3 * res * res;
    for (int col = threadIdx.x; col < kpad; col += 256) {
        float v = 0.f;
        if (col < kreal) {
            const int c = col / (patch * patch), rem = col % (patch * patch), ky = rem / patch, kx = rem % patch;
            v = src[((int64_t)c * res + py * patch + ky) * res + px * patch + kx];
        }
        io<T>::st(out + row * kpad + col, v);
    }
}

// The same for a compile-time patch size with P % 4 == 0 (ViT-B/32, B/16): four consecutive kx per thread (one 16-byte load, one 8 / 16-byte
// store) and index arithmetic by constants - the generic kernel spends its time on four run-time integer divisions per element (0.45 ms for
// the 1024 x 3 x 224 x 224 images of a micro-batch, the first kernel of the image tower).
template <typename T, int P>
__global__ __launch_bounds__(256) void im2col_p_kernel(const float* img, int res, int g, int kpad, T* out) {
    constexpr int KR = 3 * P * P;
    const int64_t row = blockIdx.x;   // b*g*g + py*g + px
    const int b = (int)(row / (g * g)), pp = (int)(row % (g * g)), py = pp / g, px = pp % g;
    const float* src = img + (int64_t)b * 3 * res * res + (int64_t)(py * P) * res + px * P;
    for (int col = threadIdx.x * 4; col < kpad; col += 1024) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (col < KR) {
            const int c = col / (P * P), rem = col % (P * P), ky = rem / P, kx = rem % P;
            v = *(const f32x4*)(src + ((int64_t)c * res + ky) * res + kx);
        }
        io<T>::st4(out + row * kpad + col, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void vit_tokens_fwd_kernel(const T* patch_out, const float* cls, const float* pos, int seq, int width, float* x) {
    const int64_t row = blockIdx.x;   // b*seq + s
    const int b = (int)(row / seq), s = (int)(row % seq);
    for (int c = threadIdx.x * 4; c < width; c += 1024) {
        f32x4 v = *(const f32x4*)(pos + (int64_t)s * width + c);
        if (s == 0) v += *(const f32x4*)(cls + c);
        else v += io<T>::ld4(patch_out + ((int64_t)b * (seq - 1) + (s - 1)) * width + c);
        *(f32x4*)(x + row * width + c) = v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void vit_tokens_bwd_copy_kernel(const float* dx, int seq, int width, T* d_patch_out) {
    const int64_t prow = blockIdx.x;  // b*(seq-1) + p
    const int b = (int)(prow / (seq - 1)), p = (int)(prow % (seq - 1));
    for (int c = threadIdx.x * 4; c < width; c += 1024)
        io<T>::st4(d_patch_out + prow * width + c, *(const f32x4*)(dx + ((int64_t)b * seq + 1 + p) * width + c));
}

// out[s][c] (+)= sum_b dx[b][s][c];  optionally extra[c] (+)= the s == 0 row (class-embedding gradient)
// A workgroup owns 64 columns of one position s: 16 threads x 16 bytes across, 16 batch groups down (b = ty, ty + 16, ...), four
// loads in flight per thread; the 16 group sums are combined through LDS in a fixed order.  grid = (ceil(width / 64), seq).
__global__ __launch_bounds__(256) void batch_sum_kernel(const float* dx, int batch, int seq, int width, float* out, float* extra, int accumulate) {
    __shared__ f32x4 sm[16][17];
    const int s = blockIdx.y;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = (blockIdx.x * 16 + tx) * 4;
    const bool live = c < width;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const float* p = dx + (int64_t)s * width + c;
        const int64_t stride = (int64_t)seq * width;
        int b = ty;
        for (; b + 48 < batch; b += 64) {
            const f32x4 v0 = *(const f32x4*)(p + b * stride), v1 = *(const f32x4*)(p + (b + 16) * stride);
            const f32x4 v2 = *(const f32x4*)(p + (b + 32) * stride), v3 = *(const f32x4*)(p + (b + 48) * stride);
            acc += v0; acc += v1; acc += v2; acc += v3;
        }
        for (; b < batch; b += 16) acc += *(const f32x4*)(p + b * stride);
    }
    sm[ty][tx] = acc;
    __syncthreads();
    if (ty != 0 || !live) return;
    acc = sm[0][tx];
#pragma unroll
    for (int k = 1; k < 16; ++k) acc += sm[k][tx];
    if (out) {
        f32x4* o = (f32x4*)(out + (int64_t)s * width + c);
        *o = accumulate ? *o + acc : acc;
    }
    if (extra && s == 0) {
        f32x4* o = (f32x4*)(extra + c);
        *o = accumulate ? *o + acc : acc;
    }
}

// one wave per row, four rows per workgroup (a workgroup per 2 KB row spent the kernel's time on dispatching 79 k workgroups: 0.5 ms for 160 MB)
__global__ __launch_bounds__(256) void text_embed_fwd_kernel(const int64_t* tokens, const float* tok_emb, const float* pos, int seq, int width,
                                                             int64_t vocab, int64_t rows, float* x) {
    const int lane = threadIdx.x & 63;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        const int s = (int)(row % seq);
        int64_t tok = tokens[row];
        tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
        for (int c = lane * 4; c < width; c += 256)
            *(f32x4*)(x + row * width + c) = *(const f32x4*)(tok_emb + tok * width + c) + *(const f32x4*)(pos + (int64_t)s * width + c);
    }
}

// Token-embedding gradient: d_tok_emb[t] += sum of the dx rows of the positions that hold token t, in a fixed order, no atomics.
// The positions arrive sorted by token (stable); a run of equal tokens belongs to ONE writer:
//   * token_scatter_kernel - one wave per sorted position, only the head of a run works; it takes the runs that lie inside one block of 64
//     sorted positions.  A lane carries all of its column quads (width / 256 of them), 16 rows in flight;
//   * token_scatter_long_kernel - one workgroup per boundary k = 64 j; the run that crosses k belongs to the FIRST boundary it crosses.
//     Start-of-text / end-of-text (one per caption) and frequent words make runs of `batch` rows: one wave walked them alone (0.56 ms at
//     batch 1024, the last kernel of the text tower's backward); here the four waves take every fourth group of 16 rows and their partial
//     sums are combined through LDS in wave order.
template <int NC>   // column quads per lane: width <= 256 NC
__device__ __forceinline__ void token_rows_sum(const float* dx, const int64_t* order, int64_t first, int64_t len, int64_t group0, int64_t gstep, int width,
                                               int lane, f32x4 (&acc)[NC]) {
#pragma unroll
    for (int q = 0; q < NC; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int64_t k = group0 * 16; k < len; k += gstep * 16) {
        const int n = (int)min((int64_t)16, len - k);
        if (n == 16) {
            f32x4 v[8][NC];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float* row = dx + order[first + k + 8 * half + j] * width;
#pragma unroll
                    for (int q = 0; q < NC; ++q) v[j][q] = (lane * 4 + 256 * q < width) ? *(const f32x4*)(row + lane * 4 + 256 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int q = 0; q < NC; ++q) acc[q] += v[j][q];
            }
        } else {
            for (int j = 0; j < n; ++j) {
                const float* row = dx + order[first + k + j] * width;
#pragma unroll
                for (int q = 0; q < NC; ++q)
                    if (lane * 4 + 256 * q < width) acc[q] += *(const f32x4*)(row + lane * 4 + 256 * q);
            }
        }
    }
}

// length of the run of `tok` that starts at sorted position i (wave-uniform), 64 positions per ballot
__device__ __forceinline__ int64_t token_run_length(const int64_t* sorted_tokens, int64_t n_sorted, int64_t i, int64_t tok, int lane) {
    int64_t len = 0;
    for (int64_t base = i;; base += 64) {
        const int64_t k = base + lane;
        const bool same = k < n_sorted && sorted_tokens[k] == tok;
        const unsigned long long m = __ballot(same);
        if (m == ~0ull) { len += 64; continue; }
        return len + __builtin_ctzll(~m);
    }
}

template <int NC>
__global__ __launch_bounds__(256) void token_scatter_kernel(const float* dx, const int64_t* sorted_tokens, const int64_t* order, int64_t n_sorted,
                                                            int width, int64_t vocab, float* d_tok_emb) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_sorted) return;
    const int64_t tok = sorted_tokens[i];
    if (i > 0 && sorted_tokens[i - 1] == tok) return;
    if (tok < 0 || tok >= vocab) return;
    const int64_t len = token_run_length(sorted_tokens, n_sorted, i, tok, lane);
    if ((i >> 6) != ((i + len - 1) >> 6)) return;          // crosses a boundary of 64: token_scatter_long_kernel's
    f32x4 acc[NC];
    token_rows_sum<NC>(dx, order, i, len, 0, 1, width, lane, acc);
#pragma unroll
    for (int q = 0; q < NC; ++q)
        if (lane * 4 + 256 * q < width) *(f32x4*)(d_tok_emb + tok * width + lane * 4 + 256 * q) += acc[q];
}

template <int NC>
__global__ __launch_bounds__(256) void token_scatter_long_kernel(const float* dx, const int64_t* sorted_tokens, const int64_t* order, int64_t n_sorted,
                                                                 int width, int64_t vocab, float* d_tok_emb) {
    __shared__ f32x4 part[3][NC][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t k = ((int64_t)blockIdx.x + 1) * 64;      // the boundary between sorted positions k - 1 and k
    if (k >= n_sorted) return;
    const int64_t tok = sorted_tokens[k];
    if (sorted_tokens[k - 1] != tok || tok < 0 || tok >= vocab) return;     // no run crosses this boundary (block-uniform)
    // the run's start: within the 64 positions in front of k, or further back - then an earlier boundary owns the run
    const bool same = sorted_tokens[k - 64 + lane] == tok;                  // k >= 64
    const unsigned long long m = __ballot(same);
    int64_t start;
    if (m == ~0ull) {                                                       // all 64 positions in front of k hold the token
        if (k > 64 && sorted_tokens[k - 65] == tok) return;                 // ... and the one in front of them: the run crosses boundary k - 64 first
        start = k - 64;
    } else {
        start = k - __builtin_clzll(~m);                                    // the highest lane that differs is the position just in front of the run
    }
    const int64_t len = (k - start) + token_run_length(sorted_tokens, n_sorted, k, tok, lane);
    f32x4 acc[NC];
    token_rows_sum<NC>(dx, order, start, len, wave, 4, width, lane, acc);   // wave w: groups of 16 rows w, w + 4, ...
    if (wave > 0) {
#pragma unroll
        for (int q = 0; q < NC; ++q) part[wave - 1][q][lane] = acc[q];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        f32x4 s = acc[q];
        s += part[0][q][lane]; s += part[1][q][lane]; s += part[2][q][lane];
        if (lane * 4 + 256 * q < width) *(f32x4*)(d_tok_emb + tok * width + lane * 4 + 256 * q) += s;
    }
}

__global__ __launch_bounds__(256) void argmax_tokens_kernel(const int64_t* tokens, int64_t batch, int seq, int32_t* eot) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= batch) return;
    int64_t best = tokens[b * seq];
    int idx = 0;
    for (int s = 1; s < seq; ++s) {
        const int64_t v = tokens[b * seq + s];
        if (v > best) { best = v; idx = s; }
    }
    eot[b] = idx;
}

// GATHER: out[b] = x[b*seq + idx[b]] ; else x[b*seq + idx[b]] = d_out[b] (dx pre-zeroed)
template <bool GATHER>
__global__ __launch_bounds__(256) void pool_kernel(const float* src, const int32_t* idx, int seq, int width, float* dst) {
    const int64_t b = blockIdx.x;
    const int64_t row = b * seq + (idx ? idx[b] : 0);
    for (int c = threadIdx.x * 4; c < width; c += 1024) {
        if (GATHER) *(f32x4*)(dst + b * width + c) = *(const f32x4*)(src + row * width + c);
        else *(f32x4*)(dst + row * width + c) = *(const f32x4*)(src + b * width + c);
    }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* src, bf16_t* dst, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) io<bf16_t>::st4(dst + i, *(const f32x4*)(src + i));
    else
        for (int64_t j = i; j < n; ++j) dst[j] = f32_to_bf16(src[j]);
}

__device__ __forceinline__ int64_t sc_cdiv_dev(int64_t a, int64_t b) { return (a + b - 1) / b; }

// dst[c][r] = bf16(src[r][c]) through a 32x33 LDS tile
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* src, int64_t rows, int64_t cols, bf16_t* dst) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t r = r0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (r < rows && c < cols) ? src[r * cols + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t c = c0 + ty + 8 * k, r = r0 + tx;
        if (c < cols && r < rows) dst[c * rows + r] = f32_to_bf16(tile[tx][ty + 8 * k]);
    }
}

// the same for a whole table of matrices in ONE launch: item i = {src, dst, rows, cols, first_block}; block -> item by binary search
__global__ __launch_bounds__(256) void transpose_cast_batch_kernel(const int64_t* table, int n_items) {
    __shared__ float tile[32][33];
    int lo = 0, hi = n_items - 1;
    const int64_t b = blockIdx.x;
    while (lo < hi) {   // last item whose first_block <= b
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 5 + 4] <= b) lo = mid; else hi = mid - 1;
    }
    const int64_t* it = table + lo * 5;
    const float* src = (const float*)it[0];
    bf16_t* dst = (bf16_t*)it[1];
    const int64_t rows = it[2], cols = it[3], local = b - it[4];
    const int64_t bx = sc_cdiv_dev(cols, 32);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t c0 = (local % bx) * 32, r0 = (local / bx) * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t r = r0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (r < rows && c < cols) ? src[r * cols + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t c = c0 + ty + 8 * k, r = r0 + tx;
        if (c < cols && r < rows) dst[c * rows + r] = f32_to_bf16(tile[tx][ty + 8 * k]);
    }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, bf16_t* shadow, int64_t n, float lr, float b1, float b2,
                                                    float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 3 < n) {
        f32x4 pv = *(f32x4*)(p + i), gv = *(const f32x4*)(g + i) * gscale, mv = *(f32x4*)(m + i), vv = *(f32x4*)(v + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pv[j] *= 1.0f - lr * wd;
            mv[j] = b1 * mv[j] + (1.0f - b1) * gv[j];
            vv[j] = b2 * vv[j] + (1.0f - b2) * gv[j] * gv[j];
            const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
            pv[j] -= (lr / bc1) * (mv[j] / denom);
        }
        *(f32x4*)(p + i) = pv; *(f32x4*)(m + i) = mv; *(f32x4*)(v + i) = vv;
        if (shadow) io<bf16_t>::st4(shadow + i, pv);
    } else {
        for (int64_t j = i; j < n; ++j) {
            float pv = p[j] * (1.0f - lr * wd);
            const float gv = g[j] * gscale;
            const float mv = b1 * m[j] + (1.0f - b1) * gv, vv = b2 * v[j] + (1.0f - b2) * gv * gv;
            pv -= (lr / bc1) * (mv / (sqrtf(vv) / bc2_sqrt + eps));
            p[j] = pv; m[j] = mv; v[j] = vv;
            if (shadow) shadow[j] = f32_to_bf16(pv);
        }
    }
}

}  // namespace

#define ST(x) ((hipStream_t)(x))

extern "C" int sc_im2col(const float* images, int64_t batch, int64_t res, int64_t patch, int64_t kpad, void* out, int dtype, void* stream) {
    SC_REQUIRE(images && out && batch > 0 && patch > 0 && res % patch == 0, SC_ERR_SHAPE, "sc_im2col: bad shape");
    const int g = (int)(res / patch), kreal = (int)(3 * patch * patch);
    SC_REQUIRE(kpad >= kreal, SC_ERR_SHAPE, "sc_im2col: kpad %lld < 3*P*P", (long long)kpad);
    const dim3 grid((unsigned)(batch * g * g));
    if ((patch == 32 || patch == 16) && kpad % 4 == 0 && res % 4 == 0 && sc_aligned(images, 16) && sc_aligned(out, 16) && (dtype == SC_BF16 || dtype == SC_F32)) {
#define IM2COL_P(T, P) hipLaunchKernelGGL((im2col_p_kernel<T, P>), grid, dim3(256), 0, ST(stream), images, (int)res, g, (int)kpad, (T*)out)
        if (dtype == SC_BF16) { if (patch == 32) IM2COL_P(bf16_t, 32); else IM2COL_P(bf16_t, 16); }
        else { if (patch == 32) IM2COL_P(float, 32); else IM2COL_P(float, 16); }
#undef IM2COL_P
        SC_CHECK_LAUNCH();
        return SC_OK;
    }
    if (dtype == SC_BF16) hipLaunchKernelGGL(im2col_kernel<bf16_t>, grid, dim3(256), 0, ST(stream), images, (int)res, (int)patch, g, kreal, (int)kpad, (bf16_t*)out);
    else if (dtype == SC_F32) hipLaunchKernelGGL(im2col_kernel<float>, grid, dim3(256), 0, ST(stream), images, (int)res, (int)patch, g, kreal, (int)kpad, (float*)out);
    else return sc_set_error(SC_ERR_DTYPE, "sc_im2col: bad dtype");
    SC_CHECK_LAUNCH();
    return SC_OK;
}

extern "C" int sc_vit_tokens_fwd(const void* patch_out, int dtype, const float* cls, const float* pos, int64_t batch, int64_t seq, int64_t width,
                                 float* x, void* stream) {
    SC_REQUIRE(patch_out && cls && pos && x && batch > 0 && seq > 1 && width % 4 == 0, SC_ERR_ARG, "sc_vit_tokens_fwd: bad argument");
    const dim3 grid((unsigned)(batch * seq));
    if (dtype == SC_BF16) hipLaunchKernelGGL(vit_tokens_fwd_kernel<bf16_t>, grid, dim3(256), 0, ST(stream), (const bf16_t*)patch_out, cls, pos, (int)seq, (int)width, x);
    else if (dtype == SC_F32) hipLaunchKernelGGL(vit_tokens_fwd_kernel<float>, grid, dim3(256), 0, ST(stream), (const float*)patch_out, cls, pos, (int)seq, (int)width, x);
    else return sc_set_error(SC_ERR_DTYPE, "sc_vit_tokens_fwd: bad dtype");
    SC_CHECK_LAUNCH();
    return SC_OK;
}

extern "C" int sc_vit_tokens_bwd(const float* dx, int64_t batch, int64_t seq, int64_t width, void* d_patch_out, int dtype, float* d_cls, float* d_pos,
                                 int accumulate, void* stream) {
    SC_REQUIRE(dx && d_patch_out && d_cls && d_pos && batch > 0 && seq > 1 && width % 4 == 0, SC_ERR_ARG, "sc_vit_tokens_bwd: bad argument");
    const dim3 grid((unsigned)(batch * (seq - 1)));
    if (dtype == SC_BF16) hipLaunchKernelGGL(vit_tokens_bwd_copy_kernel<bf16_t>, grid, dim3(256), 0, ST(stream), dx, (int)seq, (int)width, (bf16_t*)d_patch_out);
    else if (dtype == SC_F32) hipLaunchKernelGGL(vit_tokens_bwd_copy_kernel<float>, grid, dim3(256), 0, ST(stream), dx, (int)seq, (int)width, (float*)d_patch_out);
    else return sc_set_error(SC_ERR_DTYPE, "sc_vit_tokens_bwd: bad dtype");
    hipLaunchKernelGGL(batch_sum_kernel, dim3((unsigned)sc_cdiv(width, 64), (unsigned)seq), dim3(256), 0, ST(stream), dx, (int)batch, (int)seq, (int)width,
                       d_pos, d_cls, accumulate);
    SC_CHECK_LAUNCH();
    return SC_OK;
}

extern "C" int sc_text_embed_fwd(const int64_t* tokens, const float* tok_emb, const float* pos, int64_t batch, int64_t seq, int64_t width,
                                 int64_t vocab, float* x, void* stream) {
    SC_REQUIRE(tokens && tok_emb && pos && x && batch > 0 && seq > 0 && width % 4 == 0 && vocab > 0, SC_ERR_ARG, "sc_text_embed_fwd: bad argument");
    const int64_t rows = batch * seq;
    const unsigned grid = (unsigned)min((int64_t)16384, sc_cdiv(rows, 4));
    hipLaunchKernelGGL(text_embed_fwd_kernel, dim3(grid), dim3(256), 0, ST(stream), tokens, tok_emb, pos, (int)seq, (int)width, vocab, rows, x);
    SC_CHECK_LAUNCH();
    return SC_OK;
}

extern "C" int sc_text_embed_bwd(const float* dx, const int64_t* sorted_tokens, const int64_t* order, int64_t n_sorted, int64_t batch, int64_t seq,
                                 int64_t width, int64_t vocab, float* d_tok_emb, float* d_pos, int accumulate, void* stream) {
    SC_REQUIRE(dx && sorted_tokens && order && d_tok_emb && d_pos && batch > 0 && seq > 0 && width % 4 == 0, SC_ERR_ARG, "sc_text_embed_bwd: bad argument");
    SC_REQUIRE(n_sorted >= 0 && n_sorted <= batch * seq, SC_ERR_SHAPE, "sc_text_embed_bwd: n_sorted out of range");
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(d_tok_emb, 0, (size_t)vocab * width * sizeof(float), ST(stream));
        if (e != hipSuccess) return sc_set_error((int)e, "sc_text_embed_bwd: memset: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(batch_sum_kernel, dim3((unsigned)sc_cdiv(width, 64), (unsigned)seq), dim3(256), 0, ST(stream), dx, (int)batch, (int)seq, (int)width,
                       d_pos, (float*)nullptr, accumulate);
    if (n_sorted > 0) {
        SC_REQUIRE(width <= 1024, SC_ERR_SHAPE, "sc_text_embed_bwd: width %lld > 1024", (long long)width);
        const dim3 gs((unsigned)sc_cdiv(n_sorted, 4)), gl((unsigned)max((int64_t)1, (n_sorted - 1) / 64));
#define TOKEN_SCATTER(NC)                                                                                                                             \
        do {                                                                                                                                          \
            hipLaunchKernelGGL(token_scatter_kernel<NC>, gs, dim3(256), 0, ST(stream), dx, sorted_tokens, order, n_sorted, (int)width, vocab, d_tok_emb); \
            if (n_sorted > 64)                                                                                                                        \
                hipLaunchKernelGGL(token_scatter_long_kernel<NC>, gl, dim3(256), 0, ST(stream), dx, sorted_tokens, order, n_sorted, (int)width, vocab, d_tok_emb); \
        } while (0)
        if (width <= 256) TOKEN_SCATTER(1); else if (width <= 512) TOKEN_SCATTER(2); else if (width <= 768) TOKEN_SCATTER(3); else TOKEN_SCATTER(4);
#undef TOKEN_SCATTER
    }
    SC_CHECK_LAUNCH();
    return SC_OK;
}

extern "C" int sc_argmax_tokens(const int64_t* tokens, int64_t batch, int64_t seq, int32_t* eot, void* stream) {
    SC_REQUIRE(tokens && eot && batch > 0 && seq > 0, SC_ERR_ARG, "sc_argmax_tokens: bad argument");
    hipLaunchKernelGGL(argmax_tokens_kernel, dim3((unsigned)sc_cdiv(batch, 256)), dim3(256), 0, ST(stream), tokens, batch, (int)seq, eot);
    SC_CHECK_LAUNCH();
    return SC_OK;
}

extern "C" int sc_pool_gather(const float* x, const int32_t* idx, int64_t batch, int64_t seq, int64_t width, float* out, void* stream) {
    SC_REQUIRE(x && out && batch > 0 && seq > 0 && width % 4 == 0, SC_ERR_ARG, "sc_pool_gather: bad argument");
    hipLaunchKernelGGL(pool_kernel<true>, dim3((unsigned)batch), dim3(256), 0, ST(stream), x, idx, (int)seq, (int)width, out);
    SC_CHECK_LAUNCH();
    return SC_OK;
}
extern "C" int sc_pool_scatter(const float* d_out, const int32_t* idx, int64_t batch, int64_t seq, int64_t width, float* dx, void* stream) {
    SC_REQUIRE(d_out && dx && batch > 0 && seq > 0 && width % 4 == 0, SC_ERR_ARG, "sc_pool_scatter: bad argument");
    hipLaunchKernelGGL(pool_kernel<false>, dim3((unsigned)batch), dim3(256), 0, ST(stream), d_out, idx, (int)seq, (int)width, dx);
    SC_CHECK_LAUNCH();
    return SC_OK;
}

extern "C" int sc_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
    SC_REQUIRE(src && dst && n >= 0, SC_ERR_ARG, "sc_cast_f32_to_bf16: bad argument");
    if (n == 0) return SC_OK;
    SC_REQUIRE(sc_aligned(src, 16) && sc_aligned(dst, 8), SC_ERR_ALIGN, "sc_cast_f32_to_bf16: misaligned");
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)sc_cdiv(sc_cdiv(n, 4), 256)), dim3(256), 0, ST(stream), src, (bf16_t*)dst, n);
    SC_CHECK_LAUNCH();
    return SC_OK;
}
extern "C" int sc_transpose_cast_bf16_batch(const int64_t* table, int64_t n_items, int64_t total_blocks, void* stream) {
    SC_REQUIRE(table && n_items > 0 && n_items < (1 << 20) && total_blocks > 0 && total_blocks < (1ll << 31), SC_ERR_ARG, "sc_transpose_cast_bf16_batch: bad argument");
    hipLaunchKernelGGL(transpose_cast_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, ST(stream), table, (int)n_items);
    SC_CHECK_LAUNCH();
    return SC_OK;
}

extern "C" int sc_transpose_cast_bf16(const float* src, int64_t rows, int64_t cols, void* dst, void* stream) {
    SC_REQUIRE(src && dst && rows > 0 && cols > 0, SC_ERR_ARG, "sc_transpose_cast_bf16: bad argument");
    hipLaunchKernelGGL(transpose_cast_kernel, dim3((unsigned)sc_cdiv(cols, 32), (unsigned)sc_cdiv(rows, 32)), dim3(256), 0, ST(stream), src, rows, cols, (bf16_t*)dst);
    SC_CHECK_LAUNCH();
    return SC_OK;
}

extern "C" int sc_adamw_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int64_t step, float grad_scale, void* stream) {
    SC_REQUIRE(p && g && m && v && n > 0 && step >= 1, SC_ERR_ARG, "sc_adamw_step: bad argument");
    SC_REQUIRE(sc_aligned(p, 16) && sc_aligned(g, 16) && sc_aligned(m, 16) && sc_aligned(v, 16) && sc_aligned(shadow_bf16, 8), SC_ERR_ALIGN, "sc_adamw_step: misaligned");
    const float bc1 = 1.0f - (float)pow((double)beta1, (double)step);
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)sc_cdiv(sc_cdiv(n, 4), 256)), dim3(256), 0, ST(stream), p, g, m, v, (bf16_t*)shadow_bf16, n, lr, beta1, beta2,
                       eps, weight_decay, bc1, bc2_sqrt, grad_scale);
    SC_CHECK_LAUNCH();
    return SC_OK;
}
