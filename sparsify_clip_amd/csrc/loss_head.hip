// Loss head on [B,E] fp32 embeddings: symmetric InfoNCE, L_unif (pdist^2 via Gram + fused exp),
// L_align, centroid/normalise prologues, sparsify_loss, retrieval ranks.
// Reference arithmetic: sparsify_clip.py:110-132, :159-176, :186-187, :334-355, :772-773, :804, :357-416.
//
// The O(B^2 E) contractions run on the fp32 MFMA GEMM (gemm_f32.hip); the [B,B] matrices live in the caller's
// workspace (268 MB at B = 8192, against 288 GB of HBM) and are swept by row/column kernels with fixed-order
// partial sums, so every scalar is bit-stable run to run (no float atomics).
#include "common.h"
#include "gemm_epilogue.h"
#include <stdlib.h>

int sc_gemm_f32_launch(int trans_a, int trans_b, int64_t m, int64_t n, int64_t k, const float* a, int64_t lda,
                       const float* b, int64_t ldb, float* c, int64_t ldc, const EpiParams& epi, hipStream_t stream);

// fused pairwise kernels (pairwise.hip): no [B,B] matrix
bool sc_pair_supported(int64_t b, int64_t e);
size_t sc_pair_workspace_bytes(int64_t b, int64_t e);
int sc_pair_contrastive(const float* img, const float* txt, int64_t b, int64_t e, float inv_temp, float grad_scale, float* rowv, float* colv, float* diag,
                        float* d_img, float* d_txt, float** dtemp_part, int* n_dtemp, void* ws, hipStream_t st);
int sc_pair_lunif(const float* x, const float* sumsq, int64_t b, int64_t e, float t, float* rowsum, float* wx, void* ws, hipStream_t st);
int sc_pair_sparsify(const float* x, int64_t b, int64_t e, float grad_scale, float* loss_out, float* dx, float* scratch_be, void* ws, hipStream_t st);
bool sc_pair_rows_supported(int64_t b, int64_t e, int64_t row0, int64_t bm);
int sc_pair_contrastive_rows_stats(const float* img, const float* txt, int64_t b, int64_t e, int64_t row0, int64_t bm, float inv_temp, float* r_rows,
                                   float* c_rows, float* diag_rows, void* ws, hipStream_t st);
int sc_pair_contrastive_rows_grad(const float* img, const float* txt, int64_t b, int64_t e, int64_t row0, int64_t bm, float inv_temp, float grad_scale,
                                  const float* r_all, const float* c_all, float* d_img_rows, float* d_txt_rows, float* gv_out, void* ws, hipStream_t st);
int sc_pair_lunif_rows(const float* x, const float* sumsq, int64_t b, int64_t e, int64_t row0, int64_t bm, float t, float* rowsum_rows, float* wx_rows,
                       void* ws, hipStream_t st);

namespace {

// SC_LOSS_FUSED=0: the round-1 path (fp32 MFMA GEMMs on a materialised [B,B] matrix) for every shape (A/B; it stays the path of
// batches that are not a multiple of 64 and of widths that are not a multiple of 128)
bool fused_on() {
    static const bool on = [] { const char* e = sc_debug_env("SC_LOSS_FUSED"); return !(e && e[0] == '0'); }();
    return on;
}
bool use_fused(int64_t b, int64_t e) { return fused_on() && sc_pair_supported(b, e); }

constexpr int COL_CHUNKS = 64;   // row chunks of the column-statistics pass
constexpr int RED_BLOCKS = 1024; // partial sums of the matrix sweeps

struct LossWs {
    float* mat;      // [B,B]
    float* tmp;      // [B,E]
    float* rowv;     // [B]
    float* colv;     // [B]
    float* diag;     // [B]
    float* pmax;     // [COL_CHUNKS,B]
    float* psum;     // [COL_CHUNKS,B]
    float* part;     // [RED_BLOCKS * 2]
    float* scal;     // [16]
};

size_t ws_layout(int64_t b, int64_t e, void* base, LossWs* w) {
    size_t off = 0;
    auto take = [&](size_t nfloat) {
        float* p = base ? (float*)((char*)base + off) : nullptr;
        off += ((nfloat * sizeof(float) + 255) / 256) * 256;
        return p;
    };
    LossWs l;
    // fused path: the [B,B] slot is replaced by the pairwise kernels' own workspace (operand splits, partial slabs, per-tile statistics)
    l.mat = use_fused(b, e) ? take((sc_pair_workspace_bytes(b, e) + 3) / 4) : take((size_t)b * b);
    l.tmp = take((size_t)b * e);
    l.rowv = take(b);
    l.colv = take(b);
    l.diag = take(b);
    l.pmax = take((size_t)COL_CHUNKS * b);
    l.psum = take((size_t)COL_CHUNKS * b);
    l.part = take(RED_BLOCKS * 2);
    l.scal = take(16);
    if (w) *w = l;
    return off;
}

// ----------------------------------------------------------------------------- block reduce (fixed order)
__device__ __forceinline__ float block_sum_256(float v, float* sm /*[4]*/) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

// sum of n floats by one block of 256 threads: out = scale * sum  (fixed order)
__device__ __forceinline__ float serial_block_sum(const float* p, int64_t n, float* sm) {
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += p[i];
    return block_sum_256(s, sm);
}

// ----------------------------------------------------------------------------- row statistics of a [B,B] matrix
// one wave per row.  MODE 0: log-sum-exp (+ diagonal);  MODE 1: plain sum.
template <int MODE>
__global__ __launch_bounds__(256) void row_stats_kernel(const float* mat, int64_t b, float* rowv, float* diag) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= b) return;
    const float* r = mat + row * b;
    if (MODE == 0) {
        float mx = -INFINITY;
        for (int64_t j = lane; j < b; j += 64) mx = fmaxf(mx, r[j]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int64_t j = lane; j < b; j += 64) s += expf(r[j] - mx);
        s = wave_sum(s);
        if (lane == 0) {
            rowv[row] = mx + logf(s);
            diag[row] = r[row];
        }
    } else {
        float s = 0.f;
        for (int64_t j = lane; j < b; j += 64) s += r[j];
        s = wave_sum(s);
        if (lane == 0) rowv[row] = s;
    }
}

// column log-sum-exp, stage 1: thread = column, block.y = row chunk; online (max,sum)
__global__ __launch_bounds__(64) void col_lse_partial_kernel(const float* mat, int64_t b, int64_t rows_per_chunk, float* pmax, float* psum) {
    const int64_t col = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (col >= b) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = min(b, r0 + rows_per_chunk);
    float mx = -INFINITY, s = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
        const float v = mat[r * b + col];
        if (v > mx) {
            s = s * expf(mx - v) + 1.f;   // exp(-inf) = 0 on the first element
            mx = v;
        } else {
            s += expf(v - mx);
        }
    }
    pmax[(int64_t)blockIdx.y * b + col] = mx;
    psum[(int64_t)blockIdx.y * b + col] = s;
}
__global__ __launch_bounds__(256) void col_lse_final_kernel(const float* pmax, const float* psum, int64_t b, int chunks, float* colv) {
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= b) return;
    float mx = -INFINITY;
    for (int c = 0; c < chunks; ++c) mx = fmaxf(mx, pmax[(int64_t)c * b + col]);
    float s = 0.f;
    for (int c = 0; c < chunks; ++c) {
        const float pm = pmax[(int64_t)c * b + col];
        if (pm > -INFINITY) s += psum[(int64_t)c * b + col] * expf(pm - mx);
    }
    colv[col] = mx + logf(s);
}

// loss = (sum r + sum c - 2 sum diag) / (2B)
__global__ __launch_bounds__(256) void contrastive_loss_kernel(const float* rowv, const float* colv, const float* diag, int64_t b, float* loss_out) {
    __shared__ float sm[4];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < b; i += 256) s += (rowv[i] - diag[i]) + (colv[i] - diag[i]);
    s = block_sum_256(s, sm);
    if (threadIdx.x == 0) loss_out[0] = s / (2.f * (float)b);
}

// in place: G = gs * [ (exp(L - r_i) + exp(L - c_j)) / (2B) - delta_ij / B ];  part[block] = sum G*L (for d/dT)
__global__ __launch_bounds__(256) void contrastive_grad_kernel(float* mat, int64_t b, const float* rowv, const float* colv, float gs, float* part) {
    __shared__ float sm[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float inv2b = gs / (2.f * (float)b), invb = gs / (float)b;
    float acc = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < b; row += (int64_t)gridDim.x * 4) {
        float* r = mat + row * b;
        const float ri = rowv[row];
        for (int64_t j = lane; j < b; j += 64) {
            const float l = r[j];
            float g = (expf(l - ri) + expf(l - colv[j])) * inv2b;
            if (j == row) g -= invb;
            acc += g * l;
            r[j] = g;
        }
    }
    acc = block_sum_256(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
// d_temp = -(1/T) * sum_ij G_ij L_ij
__global__ __launch_bounds__(256) void dtemp_final_kernel(const float* part, int n, float inv_temp, float* d_temp) {
    __shared__ float sm[4];
    const float s = serial_block_sum(part, n, sm);
    if (threadIdx.x == 0) d_temp[0] = -inv_temp * s;
}

// ----------------------------------------------------------------------------- lunif pieces
__global__ __launch_bounds__(256) void row_sumsq_kernel(const float* x, int64_t b, int64_t e, float* out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= b) return;
    float s = 0.f;
    for (int64_t j = lane; j < e; j += 64) {
        const float v = x[row * e + j];
        s += v * v;
    }
    s = wave_sum(s);
    if (lane == 0) out[row] = s;
}
// S = sum_i s_i ; loss = log(S / (B(B-1))) ; scal[0] = S
__global__ __launch_bounds__(256) void lunif_loss_kernel(const float* rowsum, int64_t b, float* loss_out, float* scal) {
    __shared__ float sm[4];
    const float s = serial_block_sum(rowsum, b, sm);
    if (threadIdx.x == 0) {
        scal[0] = s;
        loss_out[0] = logf(s / ((float)b * (float)(b - 1)));
    }
}
// dX = gs * (-4t/S) * (s_i * x_i - (W X)_i)
__global__ __launch_bounds__(256) void lunif_grad_kernel(const float* x, const float* wx, const float* rowsum, const float* scal,
                                                         int64_t b, int64_t e, float t, float gs, float* dx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= b * e) return;
    const int64_t row = i / e;
    const float coef = gs * (-4.f * t) / scal[0];
    dx[i] = coef * (rowsum[row] * x[i] - wx[i]);
}

// ----------------------------------------------------------------------------- lalign
// one wave per row; part[block] = sum of ||x-y||^alpha over the block's rows
__global__ __launch_bounds__(256) void lalign_kernel(const float* x, const float* y, int64_t b, int64_t e, float alpha, float gs,
                                                     float* part, float* dx, float* dy) {
    __shared__ float sm[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float acc = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < b; row += (int64_t)gridDim.x * 4) {
        float ss = 0.f;
        for (int64_t j = lane; j < e; j += 64) {
            const float d = x[row * e + j] - y[row * e + j];
            ss += d * d;
        }
        ss = wave_sum(ss);
        const float nrm = sqrtf(ss);
        const float term = (alpha == 2.f) ? nrm * nrm : powf(nrm, alpha);
        if (lane == 0) acc += term;
        if (dx) {
            float coef = 0.f;   // sub-gradient 0 at zero distance, as torch.norm's backward
            if (nrm > 0.f) coef = gs * alpha * ((alpha == 2.f) ? 1.f : powf(nrm, alpha - 2.f)) / (float)b;
            for (int64_t j = lane; j < e; j += 64) {
                const float g = coef * (x[row * e + j] - y[row * e + j]);
                dx[row * e + j] = g;
                if (dy) dy[row * e + j] = -g;
            }
        }
    }
    acc = block_sum_256(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void mean_final_kernel(const float* part, int n, float inv_count, float* out) {
    __shared__ float sm[4];
    const float s = serial_block_sum(part, n, sm);
    if (threadIdx.x == 0) out[0] = s * inv_count;
}

// ----------------------------------------------------------------------------- sparsify_loss
// in place: D = G - (2I - 1); part[block] = sum D^2
__global__ __launch_bounds__(256) void sparsify_diff_kernel(float* mat, int64_t b, float* part) {
    __shared__ float sm[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float acc = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < b; row += (int64_t)gridDim.x * 4) {
        float* r = mat + row * b;
        for (int64_t j = lane; j < b; j += 64) {
            const float d = r[j] - ((j == row) ? 1.f : -1.f);
            acc += d * d;
            r[j] = d;
        }
    }
    acc = block_sum_256(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// ----------------------------------------------------------------------------- normalise / centroids
// y = x * inv, inv = 1 / max(||x||, eps)      (a := x, or a := (x + x2)/2 when x2 != null)
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* x, const float* x2, int64_t b, int64_t e, float eps, float* y, float* inv_norm) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= b) return;
    float ss = 0.f;
    for (int64_t j = lane; j < e; j += 64) {
        float v = x[row * e + j];
        if (x2) v = (v + x2[row * e + j]) / 2.0f;
        ss += v * v;
    }
    ss = wave_sum(ss);
    const float nrm = sqrtf(ss);
    const float inv = 1.f / ((eps > 0.f) ? fmaxf(nrm, eps) : nrm);
    for (int64_t j = lane; j < e; j += 64) {
        float v = x[row * e + j];
        if (x2) v = (v + x2[row * e + j]) / 2.0f;
        y[row * e + j] = v * inv;
    }
    if (lane == 0) inv_norm[row] = inv;
}
// dx = inv * (dy - y * <dy, y>);  MODE 0: write dx;  MODE 1 (centroid): d_a += dx/2, d_b += dx/2
template <int MODE>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* y, const float* inv_norm, const float* dy, int64_t b, int64_t e, float* d0, float* d1) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= b) return;
    float dot = 0.f;
    for (int64_t j = lane; j < e; j += 64) dot += dy[row * e + j] * y[row * e + j];
    dot = wave_sum(dot);
    const float inv = inv_norm[row];
    for (int64_t j = lane; j < e; j += 64) {
        const float g = inv * (dy[row * e + j] - y[row * e + j] * dot);
        if (MODE == 0) {
            d0[row * e + j] = g;
        } else {
            d0[row * e + j] += 0.5f * g;
            d1[row * e + j] += 0.5f * g;
        }
    }
}
__global__ __launch_bounds__(256) void axpy_kernel(int64_t n, float alpha, const float* x, float* y) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] += alpha * x[i];
}

// ----------------------------------------------------------------------------- retrieval ranks
// one wave per query.  forward: row i of score; backward: column i.
__global__ __launch_bounds__(256) void retrieval_kernel(const float* s, int64_t n, int32_t* rank_fwd, int32_t* rank_bwd, int32_t* top_fwd, int32_t* top_bwd) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= n) return;
    const float ref = s[q * n + q];
    for (int dir = 0; dir < 2; ++dir) {
        int cnt = 0;
        float best = -INFINITY;
        int64_t best_j = n;
        for (int64_t j = lane; j < n; j += 64) {
            const float v = dir == 0 ? s[q * n + j] : s[j * n + q];
            cnt += (v > ref) || (v == ref && j < q);
            if (v > best) { best = v; best_j = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            cnt += __shfl_xor(cnt, o, 64);
            const float ob = __shfl_xor(best, o, 64);
            const int64_t oj = __shfl_xor((long long)best_j, o, 64);
            if (ob > best || (ob == best && oj < best_j)) { best = ob; best_j = oj; }
        }
        if (lane == 0) {
            (dir == 0 ? rank_fwd : rank_bwd)[q] = cnt;
            (dir == 0 ? top_fwd : top_bwd)[q] = (int32_t)best_j;
        }
    }
}

// ----------------------------------------------------------------------------- eval geometry metrics (sparsify_clip.py:418-457, :508-528, :382-414)
// per row i: <a_i, b_i>, |a_i|^2, |b_i|^2   (one wave per row)
__global__ __launch_bounds__(256) void eval_row_stats_kernel(const float* a, const float* b_, int64_t n, int64_t e, float* rowstat /*[3][n]*/) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    float d = 0.f, na = 0.f, nb = 0.f;
    for (int64_t j = lane; j < e; j += 64) {
        const float x = a[row * e + j], y = b_[row * e + j];
        d += x * y; na += x * x; nb += y * y;
    }
    d = wave_sum(d); na = wave_sum(na); nb = wave_sum(nb);
    if (lane == 0) { rowstat[row] = d; rowstat[n + row] = na; rowstat[2 * n + row] = nb; }
}
// column sums over a chunk of rows: thread = column, blockIdx.y = chunk; part[chunk][which][e]
__global__ __launch_bounds__(256) void eval_col_partial_kernel(const float* a, const float* b_, int64_t n, int64_t e, int64_t rows_per_chunk, float* part) {
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= e) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk, r1 = min(n, r0 + rows_per_chunk);
    float sa = 0.f, sb = 0.f;
    for (int64_t r = r0; r < r1; ++r) { sa += a[r * e + col]; sb += b_[r * e + col]; }
    part[((int64_t)blockIdx.y * 2 + 0) * e + col] = sa;
    part[((int64_t)blockIdx.y * 2 + 1) * e + col] = sb;
}
// out[0] gap = |mean(a) - mean(b)|; out[1], out[2] mean off-diagonal cosine of a / b = (|sum_i x_i|^2 - sum_i |x_i|^2) / (n (n-1)) (the sum of
// all n^2 Gram entries is the squared norm of the sum vector: no [n,n] matrix); out[3] mean <a_i, b_i>;
// out[4..6] / out[7..9]: number of queries whose true match has rank < 1 / 5 / 10, forward / backward
__global__ __launch_bounds__(256) void eval_final_kernel(const float* part, int chunks, const float* rowstat, const int32_t* rank_f, const int32_t* rank_b,
                                                         int64_t n, int64_t e, float* out) {
    __shared__ float sm[4];
    float gap2 = 0.f, sa2 = 0.f, sb2 = 0.f;
    for (int64_t col = threadIdx.x; col < e; col += 256) {
        float sa = 0.f, sb = 0.f;
        for (int c = 0; c < chunks; ++c) { sa += part[((int64_t)c * 2 + 0) * e + col]; sb += part[((int64_t)c * 2 + 1) * e + col]; }
        const float d = (sa - sb) / (float)n;
        gap2 += d * d; sa2 += sa * sa; sb2 += sb * sb;
    }
    gap2 = block_sum_256(gap2, sm); sa2 = block_sum_256(sa2, sm); sb2 = block_sum_256(sb2, sm);
    float dot = 0.f, na = 0.f, nb = 0.f, cf[3] = {0.f, 0.f, 0.f}, cb[3] = {0.f, 0.f, 0.f};
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        dot += rowstat[i]; na += rowstat[n + i]; nb += rowstat[2 * n + i];
        if (rank_f) { const int r = rank_f[i]; cf[0] += r < 1; cf[1] += r < 5; cf[2] += r < 10; }
        if (rank_b) { const int r = rank_b[i]; cb[0] += r < 1; cb[1] += r < 5; cb[2] += r < 10; }
    }
    dot = block_sum_256(dot, sm); na = block_sum_256(na, sm); nb = block_sum_256(nb, sm);
    for (int k = 0; k < 3; ++k) { cf[k] = block_sum_256(cf[k], sm); cb[k] = block_sum_256(cb[k], sm); }
    if (threadIdx.x == 0) {
        const float pairs = (float)n * (float)(n - 1);
        out[0] = sqrtf(gap2);
        out[1] = (sa2 - na) / pairs;
        out[2] = (sb2 - nb) / pairs;
        out[3] = dot / (float)n;
        for (int k = 0; k < 3; ++k) { out[4 + k] = cf[k]; out[7 + k] = cb[k]; }
    }
}

inline unsigned rows4(int64_t b) { return (unsigned)sc_cdiv(b, 4); }
inline int red_blocks(int64_t b) { return (int)min((int64_t)RED_BLOCKS, sc_cdiv(b, 4)); }

int check_common(const char* who, int64_t b, int64_t e, const void* ws, size_t ws_bytes) {
    SC_REQUIRE(b >= 2 && e >= 1, SC_ERR_SHAPE, "%s: need b >= 2, e >= 1 (got %lld, %lld)", who, (long long)b, (long long)e);
    SC_REQUIRE(b <= 65536, SC_ERR_SHAPE, "%s: b too large", who);
    SC_REQUIRE(ws != nullptr, SC_ERR_WORKSPACE, "%s: workspace is null", who);
    SC_REQUIRE(sc_aligned(ws, 256), SC_ERR_ALIGN, "%s: workspace must be 256-byte aligned", who);
    SC_REQUIRE(ws_bytes >= ws_layout(b, e, nullptr, nullptr), SC_ERR_WORKSPACE, "%s: workspace too small (%zu < %zu)", who, ws_bytes,
               ws_layout(b, e, nullptr, nullptr));
    return SC_OK;
}

}  // namespace

extern "C" size_t sc_loss_workspace_bytes(int64_t b, int64_t e) {
    if (b <= 0 || e <= 0) return 0;
    return ws_layout(b, e, nullptr, nullptr);
}

extern "C" int sc_contrastive_fwd_bwd(const float* img, const float* txt, int64_t b, int64_t e, float temperature, float grad_scale,
                                      float* loss_out, float* d_img, float* d_txt, float* d_temp, void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    SC_TRY(check_common("sc_contrastive_fwd_bwd", b, e, ws, ws_bytes));
    SC_REQUIRE(img && txt && loss_out, SC_ERR_ARG, "sc_contrastive_fwd_bwd: null argument");
    SC_REQUIRE(temperature != 0.f, SC_ERR_ARG, "sc_contrastive_fwd_bwd: temperature is zero");
    SC_REQUIRE((d_img == nullptr) == (d_txt == nullptr), SC_ERR_ARG, "sc_contrastive_fwd_bwd: d_img and d_txt go together");
    LossWs w;
    ws_layout(b, e, ws, &w);
    const float inv_t = 1.0f / temperature;
    if (use_fused(b, e)) {   // tile sweeps on the bf16 matrix cores: LSE partials, then the recomputed tiles times T / I
        float* gv = nullptr;
        int ngv = 0;
        SC_TRY(sc_pair_contrastive(img, txt, b, e, inv_t, grad_scale, w.rowv, w.colv, w.diag, d_img, d_txt, (d_img && d_temp) ? &gv : nullptr, &ngv, w.mat, st));
        hipLaunchKernelGGL(contrastive_loss_kernel, dim3(1), dim3(256), 0, st, w.rowv, w.colv, w.diag, b, loss_out);
        if (gv) hipLaunchKernelGGL(dtemp_final_kernel, dim3(1), dim3(256), 0, st, gv, ngv, inv_t, d_temp);
        SC_CHECK_LAUNCH();
        return SC_OK;
    }
    // logits = I T^T / temperature   (:119-120)
    SC_TRY(sc_gemm_f32_launch(0, 1, b, b, e, img, e, txt, e, w.mat, b, epi_plain(inv_t), st));
    hipLaunchKernelGGL(row_stats_kernel<0>, dim3(rows4(b)), dim3(256), 0, st, w.mat, b, w.rowv, w.diag);
    const int chunks = (int)min((int64_t)COL_CHUNKS, sc_cdiv(b, 64));
    const int64_t rpc = sc_cdiv(b, chunks);
    hipLaunchKernelGGL(col_lse_partial_kernel, dim3((unsigned)sc_cdiv(b, 64), chunks), dim3(64), 0, st, w.mat, b, rpc, w.pmax, w.psum);
    hipLaunchKernelGGL(col_lse_final_kernel, dim3((unsigned)sc_cdiv(b, 256)), dim3(256), 0, st, w.pmax, w.psum, b, chunks, w.colv);
    hipLaunchKernelGGL(contrastive_loss_kernel, dim3(1), dim3(256), 0, st, w.rowv, w.colv, w.diag, b, loss_out);
    SC_CHECK_LAUNCH();
    if (d_img) {
        const int nb = red_blocks(b);
        hipLaunchKernelGGL(contrastive_grad_kernel, dim3(nb), dim3(256), 0, st, w.mat, b, w.rowv, w.colv, grad_scale, w.part);
        if (d_temp) hipLaunchKernelGGL(dtemp_final_kernel, dim3(1), dim3(256), 0, st, w.part, nb, inv_t, d_temp);
        SC_CHECK_LAUNCH();
        SC_TRY(sc_gemm_f32_launch(0, 0, b, e, b, w.mat, b, txt, e, d_img, e, epi_plain(inv_t), st));   // dI = G T / temp
        SC_TRY(sc_gemm_f32_launch(1, 0, b, e, b, w.mat, b, img, e, d_txt, e, epi_plain(inv_t), st));   // dT = G^T I / temp
    }
    return SC_OK;
}

extern "C" int sc_lunif_fwd_bwd(const float* x, int64_t b, int64_t e, float t, float grad_scale, float* loss_out, float* d_x, void* ws,
                                size_t ws_bytes, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    SC_TRY(check_common("sc_lunif_fwd_bwd", b, e, ws, ws_bytes));
    SC_REQUIRE(x && loss_out, SC_ERR_ARG, "sc_lunif_fwd_bwd: null argument");
    LossWs w;
    ws_layout(b, e, ws, &w);
    hipLaunchKernelGGL(row_sumsq_kernel, dim3(rows4(b)), dim3(256), 0, st, x, b, e, w.colv);
    SC_CHECK_LAUNCH();
    if (use_fused(b, e)) {   // one sweep gives the row sums of W and W X; nothing of size [B,B] exists
        SC_TRY(sc_pair_lunif(x, w.colv, b, e, t, w.rowv, w.tmp, w.mat, st));
        hipLaunchKernelGGL(lunif_loss_kernel, dim3(1), dim3(256), 0, st, w.rowv, b, loss_out, w.scal);
        if (d_x) hipLaunchKernelGGL(lunif_grad_kernel, dim3((unsigned)sc_cdiv(b * e, 256)), dim3(256), 0, st, x, w.tmp, w.rowv, w.scal, b, e, t, grad_scale, d_x);
        SC_CHECK_LAUNCH();
        return SC_OK;
    }
    // W_ij = exp(-t * max(|xi|^2 + |xj|^2 - 2 xi.xj, 0)), zero diagonal   (:161-164 via the Gram matrix)
    EpiParams ep = epi_plain();
    ep.mode = 1; ep.rowv = w.colv; ep.colv = w.colv; ep.t = t;
    SC_TRY(sc_gemm_f32_launch(0, 1, b, b, e, x, e, x, e, w.mat, b, ep, st));
    hipLaunchKernelGGL(row_stats_kernel<1>, dim3(rows4(b)), dim3(256), 0, st, w.mat, b, w.rowv, (float*)nullptr);
    hipLaunchKernelGGL(lunif_loss_kernel, dim3(1), dim3(256), 0, st, w.rowv, b, loss_out, w.scal);
    SC_CHECK_LAUNCH();
    if (d_x) {
        SC_TRY(sc_gemm_f32_launch(0, 0, b, e, b, w.mat, b, x, e, w.tmp, e, epi_plain(), st));   // W X
        hipLaunchKernelGGL(lunif_grad_kernel, dim3((unsigned)sc_cdiv(b * e, 256)), dim3(256), 0, st, x, w.tmp, w.rowv, w.scal, b, e, t,
                           grad_scale, d_x);
        SC_CHECK_LAUNCH();
    }
    return SC_OK;
}

extern "C" int sc_lalign_fwd_bwd(const float* x, const float* y, int64_t b, int64_t e, float alpha, float grad_scale, float* loss_out,
                                 float* d_x, float* d_y, void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    SC_TRY(check_common("sc_lalign_fwd_bwd", b, e, ws, ws_bytes));
    SC_REQUIRE(x && y && loss_out, SC_ERR_ARG, "sc_lalign_fwd_bwd: null argument");
    SC_REQUIRE(!(d_y && !d_x), SC_ERR_ARG, "sc_lalign_fwd_bwd: d_y without d_x");
    LossWs w;
    ws_layout(b, e, ws, &w);
    const int nb = red_blocks(b);
    hipLaunchKernelGGL(lalign_kernel, dim3(nb), dim3(256), 0, st, x, y, b, e, alpha, grad_scale, w.part, d_x, d_y);
    hipLaunchKernelGGL(mean_final_kernel, dim3(1), dim3(256), 0, st, w.part, nb, 1.0f / (float)b, loss_out);
    SC_CHECK_LAUNCH();
    return SC_OK;
}

extern "C" int sc_sparsify_fwd_bwd(const float* x, int64_t b, int64_t e, float grad_scale, float* loss_out, float* d_x, void* ws,
                                   size_t ws_bytes, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    SC_TRY(check_common("sc_sparsify_fwd_bwd", b, e, ws, ws_bytes));
    SC_REQUIRE(x && loss_out, SC_ERR_ARG, "sc_sparsify_fwd_bwd: null argument");
    LossWs w;
    ws_layout(b, e, ws, &w);
    if (use_fused(b, e)) return sc_pair_sparsify(x, b, e, grad_scale, loss_out, d_x, w.tmp, w.mat, st);
    SC_TRY(sc_gemm_f32_launch(0, 1, b, b, e, x, e, x, e, w.mat, b, epi_plain(), st));
    const int nb = red_blocks(b);
    hipLaunchKernelGGL(sparsify_diff_kernel, dim3(nb), dim3(256), 0, st, w.mat, b, w.part);
    hipLaunchKernelGGL(mean_final_kernel, dim3(1), dim3(256), 0, st, w.part, nb, 1.0f / ((float)b * (float)b), loss_out);
    SC_CHECK_LAUNCH();
    if (d_x) SC_TRY(sc_gemm_f32_launch(0, 0, b, e, b, w.mat, b, x, e, d_x, e, epi_plain(grad_scale * 4.0f / ((float)b * (float)b)), st));
    return SC_OK;
}

extern "C" int sc_l2norm_fwd(const float* x, int64_t b, int64_t e, float eps, float* y, float* inv_norm, void* stream_) {
    SC_REQUIRE(x && y && inv_norm && b > 0 && e > 0, SC_ERR_ARG, "sc_l2norm_fwd: bad argument");
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(rows4(b)), dim3(256), 0, (hipStream_t)stream_, x, (const float*)nullptr, b, e, eps, y, inv_norm);
    SC_CHECK_LAUNCH();
    return SC_OK;
}
extern "C" int sc_l2norm_bwd(const float* y, const float* inv_norm, const float* dy, int64_t b, int64_t e, float* dx, void* stream_) {
    SC_REQUIRE(y && inv_norm && dy && dx && b > 0 && e > 0, SC_ERR_ARG, "sc_l2norm_bwd: bad argument");
    hipLaunchKernelGGL(l2norm_bwd_kernel<0>, dim3(rows4(b)), dim3(256), 0, (hipStream_t)stream_, y, inv_norm, dy, b, e, dx, (float*)nullptr);
    SC_CHECK_LAUNCH();
    return SC_OK;
}
extern "C" int sc_centroid_fwd(const float* a, const float* b_, int64_t b, int64_t e, float* c, float* inv_norm, void* stream_) {
    SC_REQUIRE(a && b_ && c && inv_norm && b > 0 && e > 0, SC_ERR_ARG, "sc_centroid_fwd: bad argument");
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(rows4(b)), dim3(256), 0, (hipStream_t)stream_, a, b_, b, e, 1e-12f, c, inv_norm);
    SC_CHECK_LAUNCH();
    return SC_OK;
}
extern "C" int sc_centroid_bwd(const float* c, const float* inv_norm, const float* dc, int64_t b, int64_t e, float* d_a, float* d_b, void* stream_) {
    SC_REQUIRE(c && inv_norm && dc && d_a && d_b && b > 0 && e > 0, SC_ERR_ARG, "sc_centroid_bwd: bad argument");
    hipLaunchKernelGGL(l2norm_bwd_kernel<1>, dim3(rows4(b)), dim3(256), 0, (hipStream_t)stream_, c, inv_norm, dc, b, e, d_a, d_b);
    SC_CHECK_LAUNCH();
    return SC_OK;
}
extern "C" int sc_axpy_f32(int64_t n, float alpha, const float* x, float* y, void* stream_) {
    SC_REQUIRE(x && y && n >= 0, SC_ERR_ARG, "sc_axpy_f32: bad argument");
    if (n == 0) return SC_OK;
    hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)sc_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream_, n, alpha, x, y);
    SC_CHECK_LAUNCH();
    return SC_OK;
}
// ----------------------------------------------------------------------------- row-block (sharded) loss head
namespace {
// S = sum of the ranks' partial sums (fixed order); loss = log(S / (B(B-1))); dX_rows = gs * (-4t/S) * (s_i x_i - (W X)_i) for my rows
__global__ __launch_bounds__(256) void lunif_rows_grad_kernel(const float* x_rows, const float* wx_rows, const float* rowsum_rows, const float* s_parts,
                                                              int nparts, int64_t b, int64_t bm, int64_t e, float t, float gs, float* loss_out,
                                                              float* dx_rows) {
    float s = 0.f;
    for (int k = 0; k < nparts; ++k) s += s_parts[k];
    if (blockIdx.x == 0 && threadIdx.x == 0 && loss_out) loss_out[0] = logf(s / ((float)b * (float)(b - 1)));
    if (!dx_rows) return;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= bm * e) return;
    const int64_t row = i / e;
    const float coef = gs * (-4.f * t) / s;
    dx_rows[i] = coef * (rowsum_rows[row] * x_rows[i] - wx_rows[i]);
}
__global__ __launch_bounds__(256) void sum_to_scalar_kernel(const float* v, int64_t n, float scale, float* out) {
    __shared__ float sm[4];
    const float s = serial_block_sum(v, n, sm);
    if (threadIdx.x == 0) out[0] = s * scale;
}
int check_rows(const char* who, int64_t b, int64_t e, int64_t row0, int64_t bm, const void* ws, size_t ws_bytes) {
    SC_TRY(check_common(who, b, e, ws, ws_bytes));
    SC_REQUIRE(fused_on() && sc_pair_rows_supported(b, e, row0, bm), SC_ERR_SHAPE,
               "%s: the row-block loss head needs B %% 64 == 0, E %% 128 == 0 (E <= 1024), row0 and rows multiples of 64 (got B %lld, E %lld, rows %lld + %lld)",
               who, (long long)b, (long long)e, (long long)row0, (long long)bm);
    return SC_OK;
}
}  // namespace

extern "C" int sc_contrastive_rows_stats(const float* img, const float* txt, int64_t b, int64_t e, int64_t row0, int64_t bm, float temperature,
                                         float* stats, void* ws, size_t ws_bytes, void* stream_) {
    SC_TRY(check_rows("sc_contrastive_rows_stats", b, e, row0, bm, ws, ws_bytes));
    SC_REQUIRE(img && txt && stats && temperature != 0.f, SC_ERR_ARG, "sc_contrastive_rows_stats: bad argument");
    LossWs w;
    ws_layout(b, e, ws, &w);
    return sc_pair_contrastive_rows_stats(img, txt, b, e, row0, bm, 1.0f / temperature, stats, stats + bm, stats + 2 * bm, w.mat, (hipStream_t)stream_);
}

extern "C" int sc_contrastive_rows_grad(const float* img, const float* txt, int64_t b, int64_t e, int64_t row0, int64_t bm, float temperature,
                                        float grad_scale, const float* row_lse, const float* col_lse, const float* diag, float* loss_out,
                                        float* d_img_rows, float* d_txt_rows, float* d_temp_part, void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    SC_TRY(check_rows("sc_contrastive_rows_grad", b, e, row0, bm, ws, ws_bytes));
    SC_REQUIRE(img && txt && row_lse && col_lse && diag && loss_out && d_img_rows && d_txt_rows && temperature != 0.f, SC_ERR_ARG,
               "sc_contrastive_rows_grad: bad argument");
    LossWs w;
    ws_layout(b, e, ws, &w);
    const float inv_t = 1.0f / temperature;
    hipLaunchKernelGGL(contrastive_loss_kernel, dim3(1), dim3(256), 0, st, row_lse, col_lse, diag, b, loss_out);
    SC_TRY(sc_pair_contrastive_rows_grad(img, txt, b, e, row0, bm, inv_t, grad_scale, row_lse, col_lse, d_img_rows, d_txt_rows,
                                         d_temp_part ? w.scal : nullptr, w.mat, st));
    if (d_temp_part) hipLaunchKernelGGL(dtemp_final_kernel, dim3(1), dim3(256), 0, st, w.scal, 1, inv_t, d_temp_part);
    SC_CHECK_LAUNCH();
    return SC_OK;
}

extern "C" int sc_lunif_rows_stats(const float* x, int64_t b, int64_t e, int64_t row0, int64_t bm, float t, float* rowsum_rows, float* wx_rows,
                                   float* s_part, void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    SC_TRY(check_rows("sc_lunif_rows_stats", b, e, row0, bm, ws, ws_bytes));
    SC_REQUIRE(x && rowsum_rows && wx_rows && s_part, SC_ERR_ARG, "sc_lunif_rows_stats: null argument");
    LossWs w;
    ws_layout(b, e, ws, &w);
    hipLaunchKernelGGL(row_sumsq_kernel, dim3(rows4(b)), dim3(256), 0, st, x, b, e, w.colv);
    SC_TRY(sc_pair_lunif_rows(x, w.colv, b, e, row0, bm, t, rowsum_rows, wx_rows, w.mat, st));
    hipLaunchKernelGGL(sum_to_scalar_kernel, dim3(1), dim3(256), 0, st, rowsum_rows, bm, 1.f, s_part);
    SC_CHECK_LAUNCH();
    return SC_OK;
}

extern "C" int sc_lunif_rows_grad(const float* x_rows, int64_t b, int64_t bm, int64_t e, float t, float grad_scale, const float* s_parts, int64_t nparts,
                                  const float* rowsum_rows, const float* wx_rows, float* loss_out, float* dx_rows, void* stream_) {
    SC_REQUIRE(x_rows && s_parts && rowsum_rows && wx_rows && loss_out && b > 1 && bm > 0 && e > 0 && nparts > 0 && nparts <= 4096, SC_ERR_ARG,
               "sc_lunif_rows_grad: bad argument");
    hipLaunchKernelGGL(lunif_rows_grad_kernel, dim3((unsigned)sc_cdiv(bm * e, 256)), dim3(256), 0, (hipStream_t)stream_, x_rows, wx_rows, rowsum_rows,
                       s_parts, (int)nparts, b, bm, e, t, grad_scale, loss_out, dx_rows);
    SC_CHECK_LAUNCH();
    return SC_OK;
}

extern "C" int sc_retrieval_ranks(const float* score, int64_t n, int32_t* rank_fwd, int32_t* rank_bwd, int32_t* top1_fwd, int32_t* top1_bwd,
                                  void* stream_) {
    SC_REQUIRE(score && rank_fwd && rank_bwd && top1_fwd && top1_bwd && n > 0, SC_ERR_ARG, "sc_retrieval_ranks: bad argument");
    hipLaunchKernelGGL(retrieval_kernel, dim3(rows4(n)), dim3(256), 0, (hipStream_t)stream_, score, n, rank_fwd, rank_bwd, top1_fwd, top1_bwd);
    SC_CHECK_LAUNCH();
    return SC_OK;
}

extern "C" size_t sc_eval_metrics_workspace_bytes(int64_t n, int64_t e) {
    if (n <= 0 || e <= 0) return 0;
    return ((size_t)3 * n + (size_t)64 * 2 * e) * sizeof(float);
}
extern "C" int sc_eval_metrics(const float* img, const float* txt, int64_t n, int64_t e, const int32_t* rank_fwd, const int32_t* rank_bwd, float* out10,
                               void* ws, size_t ws_bytes, void* stream_) {
    SC_REQUIRE(img && txt && out10 && ws && n >= 2 && e >= 1, SC_ERR_ARG, "sc_eval_metrics: bad argument");
    SC_REQUIRE(ws_bytes >= sc_eval_metrics_workspace_bytes(n, e) && sc_aligned(ws, 16), SC_ERR_WORKSPACE, "sc_eval_metrics: workspace too small");
    hipStream_t st = (hipStream_t)stream_;
    float* rowstat = (float*)ws;
    float* part = rowstat + 3 * n;
    const int chunks = (int)min((int64_t)64, sc_cdiv(n, 64));
    const int64_t rpc = sc_cdiv(n, chunks);
    hipLaunchKernelGGL(eval_row_stats_kernel, dim3(rows4(n)), dim3(256), 0, st, img, txt, n, e, rowstat);
    hipLaunchKernelGGL(eval_col_partial_kernel, dim3((unsigned)sc_cdiv(e, 256), chunks), dim3(256), 0, st, img, txt, n, e, rpc, part);
    hipLaunchKernelGGL(eval_final_kernel, dim3(1), dim3(256), 0, st, part, chunks, rowstat, rank_fwd, rank_bwd, n, e, out10);
    SC_CHECK_LAUNCH();
    return SC_OK;
}
