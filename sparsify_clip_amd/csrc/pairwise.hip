// Fused global-batch loss head: the [B,B] similarity / Gram matrix is never stored.
//
// Reference arithmetic: contrastive_loss (sparsify_clip.py:110-132), lunif_loss (:159-164, pdist^2 via the Gram matrix),
// sparsify_loss (:166-176).  Each term is  (1) a B x B tile sweep  S = X Y^T  on the bf16 MFMA units,  (2) an elementwise map of
// the tile in registers (logits -> log-sum-exp partials / softmax gradient, squared distance -> exp, Gram -> residual) and, for
// the gradient,  (3) a second product  O += f(S) Z  against the rows of the other matrix - recomputing (1) instead of reading a
// stored matrix back.  Row / column statistics leave a tile as fixed-order per-tile partials (no atomics), so every result is
// bit-stable run to run.
//
// fp32 accuracy on bf16 matrix cores: every fp32 operand is split as x = hi + lo (hi = bf16(x), lo = bf16(x - hi), |x - hi - lo| <=
// 2^-17 |x|) and a product is three MFMAs, hi*hi + hi*lo + lo*hi (the dropped lo*lo term is 2^-18 relative) with fp32 accumulation:
// 3/16 of the cost of the exact fp32 MFMA path (gemm_f32.hip) at an error far below the 1e-4 bar of the golden fixtures.
//
// Tiling: a workgroup (8 waves) owns 64 rows i and sweeps column tiles of 64 rows j (a slice of them when the sweep is split
// for occupancy).  Per column tile: S[64,64] over K = E in 64-wide slices staged through LDS (hi and lo of both operands);
// f(S) is written to LDS as a bf16 hi/lo pair (the A operand of the second product); O[64,E] += P[64,64] Z_j[64,E] with every wave
// owning E/8 output columns and reading its rows of Z^T (a [E,B] copy made once per call) straight from global memory.
// At E = 512 (ViT-B/32) the gradient sweeps and the row-block statistics take the kernels further down instead (pair_grad512_kernel,
// pair_stats512_kernel): X fragments in registers, whole-width Y tiles by LDS-DMA, Z read transposed from the same LDS image.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int PT = 64;            // tile rows (i) and columns (j)
constexpr int PK = 64;            // K slice of the first product
constexpr int LDP = 72;           // LDS row stride in bf16 elements (144 B: conflict-free 16-byte row reads)
constexpr int SLICE = PT * LDP;   // elements of one staged [64][64] slice

enum { PM_CON_STATS = 0, PM_CON_GRAD = 1, PM_UNIF = 2, PM_SPARS = 3 };

struct PairParams {
    const bf16_t *xh, *xl;      // [B][E] rows i
    const bf16_t *yh, *yl;      // [B][E] rows j
    const bf16_t *zth, *ztl;    // [E][B] transposed hi / lo of the matrix the second product multiplies (gradient modes)
    int Bi, Bj, E, nsplit, jt_per_split;   // rows of X (i), rows of Y (j); the square case has Bi == Bj
    int diag_off;               // global index of row i = 0: the pair (i, j) is "on the diagonal" when i + diag_off == j (row-block form)
    float inv_temp, t;
    float coef_row, coef_col, coef_diag;   // CON_GRAD: G = coef_row exp(v - r_i) + coef_col exp(v - c_j) - coef_diag [i == j]
    const float* rowv;          // CON_GRAD: row LSE r_i (rows i);  UNIF: |x_i|^2
    const float* colv;          // CON_GRAD: column LSE c_j (rows j); UNIF: |y_j|^2
    float* opart;               // [nsplit][Bi][E] partial second products
    float *rp_m, *rp_s;         // CON_STATS: [jtiles][Bi] partial row (max, sum exp)
    float *cp_m, *cp_s;         // CON_STATS: [itiles][Bj] partial column (max, sum exp)
    float* diag;                // CON_STATS: [Bi] logits on the diagonal
    float* spart;               // UNIF: [nsplit][Bi] partial row sums of W
    float* scal_part;           // CON_GRAD: sum G v ; SPARS: sum D^2 - one float per workgroup [itiles * nsplit]
};

__device__ __forceinline__ unsigned pack_hi(float a, float b, float& ra, float& rb) {   // two bf16 hi parts + the residuals
    const unsigned u = pack2_bf16(a, b);
    ra = a - __uint_as_float(u << 16); rb = b - __uint_as_float(u & 0xffff0000u);
    return u;
}
__device__ __forceinline__ unsigned pack2(float a, float b) { return pack2_bf16(a, b); }

#define PMFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// fp32 [B][E] -> bf16 hi / lo [B][E]; with_t: also the transposed copies [E][B] (32x32 tiles through LDS)
__global__ __launch_bounds__(256) void split_kernel(const float* x, int B, int E, bf16_t* hi, bf16_t* lo, bf16_t* hi_t, bf16_t* lo_t) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        float v = 0.f;
        if (r < B && c < E) {
            v = x[(int64_t)r * E + c];
            const bf16_t h = f32_to_bf16(v);
            hi[(int64_t)r * E + c] = h;
            lo[(int64_t)r * E + c] = f32_to_bf16(v - bf16_to_f32(h));
        }
        tile[ty + 8 * k][tx] = v;
    }
    if (!hi_t) return;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (c < E && r < B) {
            const float v = tile[tx][ty + 8 * k];
            const bf16_t h = f32_to_bf16(v);
            hi_t[(int64_t)c * B + r] = h;
            lo_t[(int64_t)c * B + r] = f32_to_bf16(v - bf16_to_f32(h));
        }
    }
}

// online (max, sum exp) merge
__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
    const float mn = fmaxf(m, m2);
    if (mn == -INFINITY) { m = mn; s = 0.f; return; }
    s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
    m = mn;
}

template <int MODE, int NT2 /* E / 128: 16-column output tiles per wave of the second product */>
__global__ __launch_bounds__(512, 2) void pair_kernel(PairParams p) {
    constexpr bool GRAD = MODE != PM_CON_STATS;
    __shared__ __attribute__((aligned(16))) bf16_t smem[6 * SLICE];   // Xh Xl Yh Yl slices + Ph Pl: 55 296 B
    __shared__ float xrow[2][PT][2];     // per (wc, i): row partial (max / sum or sum)
    __shared__ float xcol[4][PT][2];     // per (wr, j): column partial
    __shared__ float xred[8];
    bf16_t* Xh = smem; bf16_t* Xl = Xh + SLICE; bf16_t* Yh = Xl + SLICE; bf16_t* Yl = Yh + SLICE; bf16_t* Ph = Yl + SLICE; bf16_t* Pl = Ph + SLICE;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int c16 = lane & 15, g = lane >> 4;
    const int it = blockIdx.x, split = blockIdx.y;
    const int i0 = it * PT;
    const int E = p.E, Bi = p.Bi, Bj = p.Bj;
    const int jt0 = split * p.jt_per_split, jt1 = min(Bj / PT, jt0 + p.jt_per_split);
    // staging: thread -> (row t >> 3, 16-byte chunk t & 7) of every [64][64] slice
    const int srow = t >> 3, schunk = (t & 7) * 8;
    const int64_t xoff = (int64_t)(i0 + srow) * E + schunk;
    const int soff = srow * LDP + schunk;
    const int i_lane = i0 + 16 * wr + c16;                       // this lane's row i in the first product
    float rowc = 0.f;
    if (MODE == PM_CON_GRAD || MODE == PM_UNIF) rowc = p.rowv[i_lane];

    f32x4 oacc[4][NT2 > 0 ? NT2 : 1];
    if (GRAD) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int n = 0; n < NT2; ++n) oacc[a][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float rsum = 0.f;        // UNIF: running row sums of W (this lane's i over this wave's columns)
    float scal = 0.f;        // CON_GRAD: sum G v; SPARS: sum D^2
    const int EC = NT2 * 16;                                   // output columns per wave of the second product
    const int nk = E / PK;

    for (int jt = jt0; jt < jt1; ++jt) {
        const int j0 = jt * PT;
        const int64_t yoff = (int64_t)(j0 + srow) * E + schunk;
        // ---- first product: S'[j][i] over K = E
        f32x4 sc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        uint4 rxh = *(const uint4*)(p.xh + xoff), rxl = *(const uint4*)(p.xl + xoff);
        uint4 ryh = *(const uint4*)(p.yh + yoff), ryl = *(const uint4*)(p.yl + yoff);
        for (int kt = 0; kt < nk; ++kt) {
            __syncthreads();                                   // the previous slice (and the previous P tile) has been consumed
            *(uint4*)(Xh + soff) = rxh; *(uint4*)(Xl + soff) = rxl; *(uint4*)(Yh + soff) = ryh; *(uint4*)(Yl + soff) = ryl;
            if (kt + 1 < nk) {
                const int ko = (kt + 1) * PK;
                rxh = *(const uint4*)(p.xh + xoff + ko); rxl = *(const uint4*)(p.xl + xoff + ko);
                ryh = *(const uint4*)(p.yh + yoff + ko); ryl = *(const uint4*)(p.yl + yoff + ko);
            }
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int fo = (16 * wr + c16) * LDP + 32 * ks + 8 * g;
                const bf16x8 ah = *(const bf16x8*)(Xh + fo), al = *(const bf16x8*)(Xl + fo);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int bo = (16 * (2 * wc + c) + c16) * LDP + 32 * ks + 8 * g;
                    const bf16x8 bh = *(const bf16x8*)(Yh + bo), bl = *(const bf16x8*)(Yl + bo);
                    sc[c] = PMFMA(bh, ah, sc[c]);
                    sc[c] = PMFMA(bl, ah, sc[c]);
                    sc[c] = PMFMA(bh, al, sc[c]);
                }
            }
        }
        // lane: row i_lane, columns j = j0 + 16 (2 wc + c) + 4 g + r
        // second product's B operand (this wave's rows e of Z^T, k = j): requested now, consumed after the map + the P barrier
        bf16x8 zh0[NT2 > 0 ? NT2 : 1], zl0[NT2 > 0 ? NT2 : 1], zh1[NT2 > 0 ? NT2 : 1], zl1[NT2 > 0 ? NT2 : 1];
        if (GRAD) {
#pragma unroll
            for (int n = 0; n < NT2; ++n) {
                const int64_t zo = (int64_t)(wave * EC + 16 * n + c16) * Bj + j0 + 8 * g;
                zh0[n] = *(const bf16x8*)(p.zth + zo);
                zl0[n] = *(const bf16x8*)(p.ztl + zo);
                zh1[n] = *(const bf16x8*)(p.zth + zo + 32);
                zl1[n] = *(const bf16x8*)(p.ztl + zo + 32);
            }
        }
        // ---- elementwise map
        if (MODE == PM_CON_STATS) {
            float m = -INFINITY;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) { sc[c][r] *= p.inv_temp; m = fmaxf(m, sc[c][r]); }
            // row partial over this wave's 32 columns
            float mr = fmaxf(m, __shfl_xor(m, 16, 64));
            mr = fmaxf(mr, __shfl_xor(mr, 32, 64));
            float sr = 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) sr += __expf(sc[c][r] - mr);
            sr += __shfl_xor(sr, 16, 64);
            sr += __shfl_xor(sr, 32, 64);
            // column partial over this wave's 16 rows: per (c, r) reduce over the 16 lanes of a lane group
            float cm[2][4], cs[2][4];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = sc[c][r];
                    x = fmaxf(x, __shfl_xor(x, 1, 64)); x = fmaxf(x, __shfl_xor(x, 2, 64));
                    x = fmaxf(x, __shfl_xor(x, 4, 64)); x = fmaxf(x, __shfl_xor(x, 8, 64));
                    float e = __expf(sc[c][r] - x);
                    e += __shfl_xor(e, 1, 64); e += __shfl_xor(e, 2, 64); e += __shfl_xor(e, 4, 64); e += __shfl_xor(e, 8, 64);
                    cm[c][r] = x; cs[c][r] = e;
                }
            __syncthreads();                                   // exchange arrays free (previous tile's readers are done)
            if (g == 0) { xrow[wc][16 * wr + c16][0] = mr; xrow[wc][16 * wr + c16][1] = sr; }
            if (c16 == 0) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int jl = 16 * (2 * wc + c) + 4 * g + r;
                        xcol[wr][jl][0] = cm[c][r]; xcol[wr][jl][1] = cs[c][r];
                    }
            }
            if (p.diag) {   // diagonal logits: i + diag_off == j
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (i_lane + p.diag_off == j0 + 16 * (2 * wc + c) + 4 * g + r) p.diag[i_lane] = sc[c][r];
            }
            __syncthreads();
            if (t < PT) {            // rows: merge the two column halves in a fixed order
                float m0 = xrow[0][t][0], s0 = xrow[0][t][1];
                lse_merge(m0, s0, xrow[1][t][0], xrow[1][t][1]);
                p.rp_m[(int64_t)jt * Bi + i0 + t] = m0; p.rp_s[(int64_t)jt * Bi + i0 + t] = s0;
            } else if (t < 2 * PT) { // columns: merge the four row tiles
                const int jl = t - PT;
                float m0 = xcol[0][jl][0], s0 = xcol[0][jl][1];
                lse_merge(m0, s0, xcol[1][jl][0], xcol[1][jl][1]);
                lse_merge(m0, s0, xcol[2][jl][0], xcol[2][jl][1]);
                lse_merge(m0, s0, xcol[3][jl][0], xcol[3][jl][1]);
                p.cp_m[(int64_t)it * Bj + j0 + jl] = m0; p.cp_s[(int64_t)it * Bj + j0 + jl] = s0;
            }
            continue;
        }
        // gradient modes: P = f(S) as a bf16 hi / lo pair in LDS, row-major [i][j]
        f32x4 pv[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x4 colc = {0.f, 0.f, 0.f, 0.f};
            if (MODE == PM_CON_GRAD || MODE == PM_UNIF) colc = *(const f32x4*)(p.colv + j0 + 16 * (2 * wc + c) + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool dg = i_lane + p.diag_off == j0 + 16 * (2 * wc + c) + 4 * g + r;
                const float s = sc[c][r];
                float v;
                if (MODE == PM_CON_GRAD) {
                    const float l = s * p.inv_temp;
                    v = p.coef_row * __expf(l - rowc) + p.coef_col * __expf(l - colc[r]) - (dg ? p.coef_diag : 0.f);
                    scal += v * l;
                } else if (MODE == PM_UNIF) {
                    v = dg ? 0.f : __expf(-p.t * fmaxf(rowc + colc[r] - 2.f * s, 0.f));
                    rsum += v;
                } else {
                    v = s - (dg ? 1.f : -1.f);
                    scal += v * v;
                }
                pv[c][r] = v;
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float r0, r1, r2, r3;
            uint2 h, l;
            h.x = pack_hi(pv[c][0], pv[c][1], r0, r1); h.y = pack_hi(pv[c][2], pv[c][3], r2, r3);
            l.x = pack2(r0, r1); l.y = pack2(r2, r3);
            const int po = (16 * wr + c16) * LDP + 16 * (2 * wc + c) + 4 * g;
            *(uint2*)(Ph + po) = h; *(uint2*)(Pl + po) = l;
        }
        __syncthreads();                                       // P complete
        // ---- second product: O'[e][i] += Z^T[e][j] P[i][j]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int ao = (16 * a + c16) * LDP + 32 * ks + 8 * g;
                const bf16x8 ph = *(const bf16x8*)(Ph + ao), pl = *(const bf16x8*)(Pl + ao);
#pragma unroll
                for (int n = 0; n < NT2; ++n) {
                    const bf16x8 zh = ks == 0 ? zh0[n] : zh1[n], zl = ks == 0 ? zl0[n] : zl1[n];
                    oacc[a][n] = PMFMA(zh, ph, oacc[a][n]);
                    oacc[a][n] = PMFMA(zl, ph, oacc[a][n]);
                    oacc[a][n] = PMFMA(zh, pl, oacc[a][n]);
                }
            }
        }
    }
    if (!GRAD) return;
    // ---- outputs of the sweep: O partial, row-sum partial, scalar partial
    float* od = p.opart + ((int64_t)split * Bi + i0) * E + wave * EC;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int n = 0; n < NT2; ++n) *(f32x4*)(od + (int64_t)(16 * a + c16) * E + 16 * n + 4 * g) = oacc[a][n];   // lane: row i = 16 a + c16, columns 16 n + 4 g ..
    if (MODE == PM_UNIF) {
        rsum += __shfl_xor(rsum, 16, 64);
        rsum += __shfl_xor(rsum, 32, 64);
        __syncthreads();
        if (g == 0) xrow[wc][16 * wr + c16][0] = rsum;
        __syncthreads();
        if (t < PT) p.spart[(int64_t)split * Bi + i0 + t] = xrow[0][t][0] + xrow[1][t][0];
    } else {
        scal = wave_sum(scal);
        __syncthreads();
        if (lane == 0) xred[wave] = scal;
        __syncthreads();
        if (t == 0) p.scal_part[it * p.nsplit + split] = ((xred[0] + xred[1]) + (xred[2] + xred[3])) + ((xred[4] + xred[5]) + (xred[6] + xred[7]));
    }
}

// ------------------------------------------------------------------------------------------------ gradient sweeps at E = 512
// The kernel above moves 384 KiB from L2 per 64 x 64 tile pair (the X slices again for every column tile, the Y slices, Z^T fragments from
// global) - twice what a CU can pull in the time its MFMAs need - behind 18 barriers.  Here, for the embedding width of ViT-B/32:
//   * the wave's 16 rows of X live in registers as MFMA fragments (hi and lo: 128 registers), loaded once per sweep;
//   * a column tile is 32 rows of Y over the WHOLE width, brought by LDS-DMA (one instruction per 1 KiB row, rows padded to 1056 B: the
//     16-byte row reads of the first product and the transposed reads of the second are both conflict-free) - 64 KiB per 64 x 32 pair;
//   * Z = Y in every mode of the loss head, so the second product reads its A operand (Z^T) out of the SAME image with
//     ds_read_b64_tr_b16; its k-slots are rows {4g .. 4g+3, 16+4g .. 16+4g+3}, and the P fragment is read from its LDS tile in that order;
//   * three barriers per column tile.  78 KiB of LDS, two workgroups per CU.
constexpr int GE = 512, GJ = 32, LDY = GE + 16, LDQ = 40;   // LDS row strides in elements: Y 1056 B, P 80 B
typedef __attribute__((address_space(3))) bf16x4* pair_ltr_t;
typedef const __attribute__((address_space(1))) void* pair_gptr_t;
typedef __attribute__((address_space(3))) void* pair_lptr_t;

// (A second Y buffer - the next column tile arriving under this one's arithmetic, 146 KiB, one workgroup per CU - measured the same as this
// form, where two workgroups per CU cover each other's waits: 2.30 vs 2.25 ms for the experiment_6 stack at B = 8192.)
template <int MODE>
__global__ __launch_bounds__(512, 2) void pair_grad512_kernel(PairParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t ysm[2 * GJ * LDY];
    __shared__ __attribute__((aligned(16))) bf16_t psm[2 * PT * LDQ];
    __shared__ float xrow[2][PT];
    __shared__ float xred[8];
    bf16_t* Ph = psm; bf16_t* Pl = psm + PT * LDQ;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int c16 = lane & 15, g = lane >> 4;
    const int it = blockIdx.x, split = blockIdx.y;
    const int i0 = it * PT;
    const int Bi = p.Bi, Bj = p.Bj;
    const int j_begin = split * p.jt_per_split * PT, j_end = min(Bj, j_begin + p.jt_per_split * PT);
    const int i_lane = i0 + 16 * wr + c16;
    float rowc = 0.f;
    if (MODE == PM_CON_GRAD || MODE == PM_UNIF) rowc = p.rowv[i_lane];
    bf16x8 xh[GE / 32], xl[GE / 32];
#pragma unroll
    for (int ks = 0; ks < GE / 32; ++ks) {
        xh[ks] = *(const bf16x8*)(p.xh + (int64_t)i_lane * GE + 32 * ks + 8 * g);
        xl[ks] = *(const bf16x8*)(p.xl + (int64_t)i_lane * GE + 32 * ks + 8 * g);
    }
    f32x4 oacc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int n = 0; n < 4; ++n) oacc[a][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float rsum = 0.f, scal = 0.f;

    auto bring = [&](int j0, bf16_t* yh, bf16_t* yl) {       // wave w: rows 4 w .. 4 w + 3, hi and lo: one 1 KiB row per LDS-DMA instruction
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 4 * wave + q;
            __builtin_amdgcn_global_load_lds((pair_gptr_t)(p.yh + (int64_t)(j0 + r) * GE + lane * 8), (pair_lptr_t)(yh + r * LDY), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((pair_gptr_t)(p.yl + (int64_t)(j0 + r) * GE + lane * 8), (pair_lptr_t)(yl + r * LDY), 16, 0, 0);
        }
    };
    bf16_t* Yh = ysm; bf16_t* Yl = ysm + GJ * LDY;
    for (int j0 = j_begin; j0 < j_end; j0 += GJ) {
        f32x4 colc = {0.f, 0.f, 0.f, 0.f};
        if (MODE == PM_CON_GRAD || MODE == PM_UNIF) colc = *(const f32x4*)(p.colv + j0 + 16 * wc + 4 * g);
        __syncthreads();                                       // the previous tile's readers of Y and P are done
        bring(j0, Yh, Yl);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- first product: S'[j][i], this wave's 16 columns j0 + 16 wc .. over K = E; one accumulator per term (hi hi, lo hi, hi lo)
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0;   // (an MFMA that accumulates into the previous one's result waits for it: rotate)
#pragma unroll
        for (int ks = 0; ks < GE / 32; ++ks) {
            const int fo = (16 * wc + c16) * LDY + 32 * ks + 8 * g;
            const bf16x8 bh = *(const bf16x8*)(Yh + fo), bl = *(const bf16x8*)(Yl + fo);
            s0 = PMFMA(bh, xh[ks], s0);
            s1 = PMFMA(bl, xh[ks], s1);
            s2 = PMFMA(bh, xl[ks], s2);
        }
        // ---- elementwise map: lane = row i_lane, columns j = j0 + 16 wc + 4 g + r
        f32x4 pv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool dg = i_lane + p.diag_off == j0 + 16 * wc + 4 * g + r;
            const float sv = s0[r] + (s1[r] + s2[r]);
            float v;
            if (MODE == PM_CON_GRAD) {
                const float l = sv * p.inv_temp;
                v = p.coef_row * __expf(l - rowc) + p.coef_col * __expf(l - colc[r]) - (dg ? p.coef_diag : 0.f);
                scal += v * l;
            } else if (MODE == PM_UNIF) {
                v = dg ? 0.f : __expf(-p.t * fmaxf(rowc + colc[r] - 2.f * sv, 0.f));
                rsum += v;
            } else {
                v = sv - (dg ? 1.f : -1.f);
                scal += v * v;
            }
            pv[r] = v;
        }
        {
            float r0, r1, r2, r3;
            uint2 h, l;
            h.x = pack_hi(pv[0], pv[1], r0, r1); h.y = pack_hi(pv[2], pv[3], r2, r3);
            l.x = pack2(r0, r1); l.y = pack2(r2, r3);
            const int po = (16 * wr + c16) * LDQ + 16 * wc + 4 * g;
            *(uint2*)(Ph + po) = h; *(uint2*)(Pl + po) = l;
        }
        __syncthreads();                                       // P complete
        // ---- second product: O'[e][i] += Z^T[e][j] P[i][j], this wave's output columns e = 64 wave ..; the P fragments of the four row tiles
        // are read once and held (every wave reads the whole P tile), the Z^T fragments once per 16 output columns
        bf16x8 ph[4], pl[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int ao = (16 * a + c16) * LDQ + 4 * g;   // row i = 16 a + c16: columns 4 g .. 4 g + 3 and 16 + 4 g .. (the k-slots of the transposed read)
            const bf16x4 ph0 = *(const bf16x4*)(Ph + ao), ph1 = *(const bf16x4*)(Ph + ao + 16);
            const bf16x4 pl0 = *(const bf16x4*)(Pl + ao), pl1 = *(const bf16x4*)(Pl + ao + 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) { ph[a][e] = ph0[e]; ph[a][4 + e] = ph1[e]; pl[a][e] = pl0[e]; pl[a][4 + e] = pl1[e]; }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int zo = (4 * g + (c16 >> 2)) * LDY + 64 * wave + 16 * n + 4 * (c16 & 3);   // this lane's 8-byte piece of the transposed read
            const bf16x4 zh0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pair_ltr_t)(Yh + zo)), zh1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pair_ltr_t)(Yh + zo + 16 * LDY));
            const bf16x4 zl0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pair_ltr_t)(Yl + zo)), zl1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pair_ltr_t)(Yl + zo + 16 * LDY));
            bf16x8 zh, zl;
#pragma unroll
            for (int e = 0; e < 4; ++e) { zh[e] = zh0[e]; zh[4 + e] = zh1[e]; zl[e] = zl0[e]; zl[4 + e] = zl1[e]; }
            // the three terms of an output tile four MFMAs apart (an MFMA that accumulates into the previous one's result waits for it)
#pragma unroll
            for (int a = 0; a < 4; ++a) oacc[a][n] = PMFMA(zh, ph[a], oacc[a][n]);
#pragma unroll
            for (int a = 0; a < 4; ++a) oacc[a][n] = PMFMA(zl, ph[a], oacc[a][n]);
#pragma unroll
            for (int a = 0; a < 4; ++a) oacc[a][n] = PMFMA(zh, pl[a], oacc[a][n]);
        }
    }
    // ---- outputs of the sweep: O partial, row-sum partial, scalar partial (as pair_kernel)
    float* od = p.opart + ((int64_t)split * Bi + i0) * GE + wave * 64;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int n = 0; n < 4; ++n) *(f32x4*)(od + (int64_t)(16 * a + c16) * GE + 16 * n + 4 * g) = oacc[a][n];
    if (MODE == PM_UNIF) {
        rsum += __shfl_xor(rsum, 16, 64);
        rsum += __shfl_xor(rsum, 32, 64);
        __syncthreads();
        if (g == 0) xrow[wc][16 * wr + c16] = rsum;
        __syncthreads();
        if (t < PT) p.spart[(int64_t)split * Bi + i0 + t] = xrow[0][t] + xrow[1][t];
    } else {
        scal = wave_sum(scal);
        __syncthreads();
        if (lane == 0) xred[wave] = scal;
        __syncthreads();
        if (t == 0) p.scal_part[it * p.nsplit + split] = ((xred[0] + xred[1]) + (xred[2] + xred[3])) + ((xred[4] + xred[5]) + (xred[6] + xred[7]));
    }
}

// Row statistics of the logits at E = 512 (same operand handling as pair_grad512_kernel, first product only): every lane carries the
// running (max, sum exp) of its row over its columns through the whole sweep, so a workgroup leaves ONE partial per row and split -
// rp_m / rp_s [nsplit][Bi] - and no column statistics: the column LSE of the square problem is the row LSE of the transposed one (a
// second sweep of half a gradient sweep's MFMA work, against the 32 cross-lane exchanges per column tile of pair_kernel<PM_CON_STATS>).
__global__ __launch_bounds__(512, 2) void pair_stats512_kernel(PairParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t ysm[2 * GJ * LDY];
    __shared__ float xm[2][PT], xs[2][PT];
    bf16_t* Yh = ysm; bf16_t* Yl = ysm + GJ * LDY;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int c16 = lane & 15, g = lane >> 4;
    const int it = blockIdx.x, split = blockIdx.y;
    const int i0 = it * PT;
    const int Bi = p.Bi, Bj = p.Bj;
    const int j_begin = split * p.jt_per_split * PT, j_end = min(Bj, j_begin + p.jt_per_split * PT);
    const int i_lane = i0 + 16 * wr + c16;
    bf16x8 xh[GE / 32], xl[GE / 32];
#pragma unroll
    for (int ks = 0; ks < GE / 32; ++ks) {
        xh[ks] = *(const bf16x8*)(p.xh + (int64_t)i_lane * GE + 32 * ks + 8 * g);
        xl[ks] = *(const bf16x8*)(p.xl + (int64_t)i_lane * GE + 32 * ks + 8 * g);
    }
    float m = -INFINITY, sum = 0.f;
    for (int j0 = j_begin; j0 < j_end; j0 += GJ) {
        __syncthreads();                                       // the previous tile's readers are done
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 4 * wave + q;
            __builtin_amdgcn_global_load_lds((pair_gptr_t)(p.yh + (int64_t)(j0 + r) * GE + lane * 8), (pair_lptr_t)(Yh + r * LDY), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((pair_gptr_t)(p.yl + (int64_t)(j0 + r) * GE + lane * 8), (pair_lptr_t)(Yl + r * LDY), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0;
#pragma unroll
        for (int ks = 0; ks < GE / 32; ++ks) {
            const int fo = (16 * wc + c16) * LDY + 32 * ks + 8 * g;
            const bf16x8 bh = *(const bf16x8*)(Yh + fo), bl = *(const bf16x8*)(Yl + fo);
            s0 = PMFMA(bh, xh[ks], s0);
            s1 = PMFMA(bl, xh[ks], s1);
            s2 = PMFMA(bh, xl[ks], s2);
        }
        float v[4], mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = (s0[r] + (s1[r] + s2[r])) * p.inv_temp;
            mt = fmaxf(mt, v[r]);
            if (p.diag && i_lane + p.diag_off == j0 + 16 * wc + 4 * g + r) p.diag[i_lane] = v[r];
        }
        const float mn = fmaxf(m, mt);                         // finite: mt is
        sum = sum * __expf(m - mn) + ((__expf(v[0] - mn) + __expf(v[1] - mn)) + (__expf(v[2] - mn) + __expf(v[3] - mn)));
        m = mn;
    }
    // the four lane groups of a row, then the two column halves, in a fixed order
    lse_merge(m, sum, __shfl_xor(m, 16, 64), __shfl_xor(sum, 16, 64));
    lse_merge(m, sum, __shfl_xor(m, 32, 64), __shfl_xor(sum, 32, 64));
    if (g == 0) { xm[wc][16 * wr + c16] = m; xs[wc][16 * wr + c16] = sum; }
    __syncthreads();
    if (t < PT) {
        float m0 = xm[0][t], s0 = xs[0][t];
        lse_merge(m0, s0, xm[1][t], xs[1][t]);
        p.rp_m[(int64_t)split * Bi + i0 + t] = m0; p.rp_s[(int64_t)split * Bi + i0 + t] = s0;
    }
}

// ------------------------------------------------------------------------------------------------ finalisation kernels
// LSE over the partials of one row / column: out[i] = log sum_k s_k exp(m_k).  32 outputs per workgroup, the partials of an output dealt
// to 8 threads (merged on-line, then in slice order through LDS): one thread per output walked 128 partials 32 KiB apart on 32 workgroups
// (95 us at B = 8192, twice per contrastive term).
constexpr int LSE_OUT = 32;   // outputs per workgroup of lse_final_kernel
__global__ __launch_bounds__(256) void lse_final_kernel(const float* pm, const float* ps, int parts, int B, float* out) {
    __shared__ float sm_m[8][LSE_OUT], sm_s[8][LSE_OUT];
    const int tx = threadIdx.x & (LSE_OUT - 1), ty = threadIdx.x / LSE_OUT;
    const int i = blockIdx.x * LSE_OUT + tx;
    float m = -INFINITY, s = 0.f;
    if (i < B)
        for (int k = ty; k < parts; k += 8) lse_merge(m, s, pm[(int64_t)k * B + i], ps[(int64_t)k * B + i]);
    sm_m[ty][tx] = m; sm_s[ty][tx] = s;
    __syncthreads();
    if (ty != 0 || i >= B) return;
#pragma unroll
    for (int k = 1; k < 8; ++k) lse_merge(m, s, sm_m[k][tx], sm_s[k][tx]);
    out[i] = m + logf(s);
}
// out = scale * sum_s part[s]   (fixed order), n elements
__global__ __launch_bounds__(256) void sum_splits_kernel(const float* part, int nsplit, int64_t n, float scale, float* out) {
    const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= n) return;
    f32x4 s = *(const f32x4*)(part + i4);
    for (int k = 1; k < nsplit; ++k) s += *(const f32x4*)(part + (int64_t)k * n + i4);
    *(f32x4*)(out + i4) = s * scale;
}
// scalar: out[0] = scale * sum of n floats (one block, fixed order)
__global__ __launch_bounds__(256) void sum_scalar_kernel(const float* part, int n, float scale, float* out) {
    __shared__ float sm[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = ((sm[0] + sm[1]) + (sm[2] + sm[3])) * scale;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ host side (called from loss_head.hip)
struct PairWs {      // carved from the caller's workspace
    bf16_t *xh, *xl, *yh, *yl, *xth, *xtl, *yth, *ytl;
    float *opart, *rp_m, *rp_s, *cp_m, *cp_s, *spart, *scal_part;
};

static int pair_nsplit(int64_t b) {
    const int64_t it = b / PT;
    static const int wgs = [] { const char* e = sc_debug_env("SC_PAIR_WGS"); return e && atoi(e) > 0 ? atoi(e) : 512; }();   // two workgroups per CU hide each other's barriers
    int64_t s = wgs / it;
    if (s < 1) s = 1;
    if (s > it) s = it;
    return (int)s;
}

size_t sc_pair_workspace_bytes(int64_t b, int64_t e) {
    const int64_t it = b / PT, ns = pair_nsplit(b);
    size_t n = 0;
    auto add = [&](size_t bytes) { n += ((bytes + 255) / 256) * 256; };
    for (int k = 0; k < 8; ++k) add((size_t)b * e * 2);
    add((size_t)ns * b * e * 4);
    for (int k = 0; k < 4; ++k) add((size_t)it * b * 4);
    add((size_t)ns * b * 4);
    add((size_t)it * ns * 4 + 64);
    return n;
}

static void pair_carve(void* ws, int64_t b, int64_t e, PairWs& w) {
    const int64_t it = b / PT, ns = pair_nsplit(b);
    char* c = (char*)ws;
    auto take = [&](size_t bytes) { char* r = c; c += ((bytes + 255) / 256) * 256; return r; };
    bf16_t** hs[8] = {&w.xh, &w.xl, &w.yh, &w.yl, &w.xth, &w.xtl, &w.yth, &w.ytl};
    for (auto h : hs) *h = (bf16_t*)take((size_t)b * e * 2);
    w.opart = (float*)take((size_t)ns * b * e * 4);
    w.rp_m = (float*)take((size_t)it * b * 4); w.rp_s = (float*)take((size_t)it * b * 4);
    w.cp_m = (float*)take((size_t)it * b * 4); w.cp_s = (float*)take((size_t)it * b * 4);
    w.spart = (float*)take((size_t)ns * b * 4);
    w.scal_part = (float*)take((size_t)it * ns * 4 + 64);
}

bool sc_pair_supported(int64_t b, int64_t e) { return b >= 2 * PT && b % PT == 0 && e % 128 == 0 && e >= 128 && e <= 1024 && b <= 65536; }

static bool pair_wide_off() {
    static const bool off = [] { const char* e = sc_debug_env("SC_PAIR_OLD"); return e && e[0] == '1'; }();   // =1: the sliced kernel at every width (A/B)
    return off;
}
static bool pair_stats512(const PairParams& p) { return p.E == GE && !pair_wide_off(); }

template <int MODE>
static int pair_launch(const PairParams& p, hipStream_t st) {
    const dim3 grid((unsigned)(p.Bi / PT), (unsigned)p.nsplit);
    if constexpr (MODE != PM_CON_STATS) {
        if (p.E == GE && !pair_wide_off()) {   // column tiles of 32 rows: Bj % 64 == 0 (sc_pair_supported) covers it
            hipLaunchKernelGGL((pair_grad512_kernel<MODE>), grid, dim3(512), 0, st, p);
            SC_CHECK_LAUNCH();
            return SC_OK;
        }
    }
    switch (p.E / 128) {
#define CASE(N) case N: hipLaunchKernelGGL((pair_kernel<MODE, N>), grid, dim3(512), 0, st, p); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
        default: return sc_set_error(SC_ERR_SHAPE, "pairwise kernel: unsupported width %d", p.E);
    }
    SC_CHECK_LAUNCH();
    return SC_OK;
}

static void split_launch(const float* x, int64_t b, int64_t e, bf16_t* hi, bf16_t* lo, bf16_t* hi_t, bf16_t* lo_t, hipStream_t st) {
    hipLaunchKernelGGL(split_kernel, dim3((unsigned)sc_cdiv(e, 32), (unsigned)sc_cdiv(b, 32)), dim3(256), 0, st, x, (int)b, (int)e, hi, lo, hi_t, lo_t);
}

// contrastive: row / column LSE + diagonal into rowv / colv / diag ([B] each); with d_img: both gradients and, if dtemp_part, the
// partial sums of G v (count returned in *n_dtemp)
int sc_pair_contrastive(const float* img, const float* txt, int64_t b, int64_t e, float inv_temp, float grad_scale, float* rowv, float* colv, float* diag,
                        float* d_img, float* d_txt, float** dtemp_part, int* n_dtemp, void* ws, hipStream_t st) {
    PairWs w;
    pair_carve(ws, b, e, w);
    const bool grad = d_img != nullptr;
    split_launch(img, b, e, w.xh, w.xl, grad ? w.xth : nullptr, grad ? w.xtl : nullptr, st);
    split_launch(txt, b, e, w.yh, w.yl, grad ? w.yth : nullptr, grad ? w.ytl : nullptr, st);
    PairParams p = {};
    p.xh = w.xh; p.xl = w.xl; p.yh = w.yh; p.yl = w.yl; p.Bi = p.Bj = (int)b; p.E = (int)e; p.inv_temp = inv_temp;
    p.nsplit = pair_nsplit(b); p.jt_per_split = (int)sc_cdiv(b / PT, p.nsplit);
    p.rp_m = w.rp_m; p.rp_s = w.rp_s; p.cp_m = w.cp_m; p.cp_s = w.cp_s; p.diag = diag;
    static const bool two_sweeps = [] { const char* e = sc_debug_env("SC_PAIR_STATS2"); return e && e[0] == '1'; }();   // (A/B: 2.18 vs 2.13 ms for the experiment_6 stack)
    if (two_sweeps && pair_stats512(p)) {   // rows-only statistics: the column LSE is the row LSE of the transposed problem
        const dim3 grid((unsigned)(b / PT), (unsigned)p.nsplit);
        hipLaunchKernelGGL(pair_stats512_kernel, grid, dim3(512), 0, st, p);
        hipLaunchKernelGGL(lse_final_kernel, dim3((unsigned)sc_cdiv(b, LSE_OUT)), dim3(256), 0, st, w.rp_m, w.rp_s, p.nsplit, (int)b, rowv);
        PairParams q = p;
        q.xh = w.yh; q.xl = w.yl; q.yh = w.xh; q.yl = w.xl; q.diag = nullptr;
        hipLaunchKernelGGL(pair_stats512_kernel, grid, dim3(512), 0, st, q);
        hipLaunchKernelGGL(lse_final_kernel, dim3((unsigned)sc_cdiv(b, LSE_OUT)), dim3(256), 0, st, w.rp_m, w.rp_s, p.nsplit, (int)b, colv);
    } else {
        SC_TRY(pair_launch<PM_CON_STATS>(p, st));
        const int parts = (int)(b / PT);
        hipLaunchKernelGGL(lse_final_kernel, dim3((unsigned)sc_cdiv(b, LSE_OUT)), dim3(256), 0, st, w.rp_m, w.rp_s, parts, (int)b, rowv);
        hipLaunchKernelGGL(lse_final_kernel, dim3((unsigned)sc_cdiv(b, LSE_OUT)), dim3(256), 0, st, w.cp_m, w.cp_s, parts, (int)b, colv);
    }
    SC_CHECK_LAUNCH();
    if (!grad) return SC_OK;
    // G = gs [ (exp(L - r_i) + exp(L - c_j)) / (2B) - delta_ij / B ];  dI = G T / temp,  dT = G^T I / temp
    p.coef_row = p.coef_col = grad_scale / (2.f * (float)b); p.coef_diag = grad_scale / (float)b;
    p.opart = w.opart; p.scal_part = w.scal_part;
    const int64_t n = b * e;
    const unsigned rb = (unsigned)sc_cdiv(n / 4, 256);
    p.rowv = rowv; p.colv = colv; p.zth = w.yth; p.ztl = w.ytl;                     // rows = images, columns = texts, Z = T
    SC_TRY(pair_launch<PM_CON_GRAD>(p, st));
    hipLaunchKernelGGL(sum_splits_kernel, dim3(rb), dim3(256), 0, st, w.opart, p.nsplit, n, inv_temp, d_img);
    if (dtemp_part) {   // sum G v, taken from the first sweep: hand the partials over before the second sweep overwrites them
        float* keep = w.spart;   // free in the contrastive term
        hipLaunchKernelGGL(sum_scalar_kernel, dim3(1), dim3(256), 0, st, w.scal_part, (int)(b / PT) * p.nsplit, 1.f, keep);
        *dtemp_part = keep; *n_dtemp = 1;
    }
    PairParams q = p;                                                                // transposed problem: rows = texts
    q.xh = w.yh; q.xl = w.yl; q.yh = w.xh; q.yl = w.xl; q.rowv = colv; q.colv = rowv; q.zth = w.xth; q.ztl = w.xtl;
    SC_TRY(pair_launch<PM_CON_GRAD>(q, st));
    hipLaunchKernelGGL(sum_splits_kernel, dim3(rb), dim3(256), 0, st, w.opart, p.nsplit, n, inv_temp, d_txt);
    SC_CHECK_LAUNCH();
    return SC_OK;
}

// lunif: W = exp(-t max(|xi|^2 + |xj|^2 - 2 xi.xj, 0)) off the diagonal: row sums s_i -> rowsum[B], W X -> wx[B,E] (both final)
int sc_pair_lunif(const float* x, const float* sumsq, int64_t b, int64_t e, float t, float* rowsum, float* wx, void* ws, hipStream_t st) {
    PairWs w;
    pair_carve(ws, b, e, w);
    split_launch(x, b, e, w.xh, w.xl, w.xth, w.xtl, st);
    PairParams p = {};
    p.xh = w.xh; p.xl = w.xl; p.yh = w.xh; p.yl = w.xl; p.zth = w.xth; p.ztl = w.xtl; p.Bi = p.Bj = (int)b; p.E = (int)e; p.t = t;
    p.nsplit = pair_nsplit(b); p.jt_per_split = (int)sc_cdiv(b / PT, p.nsplit);
    p.rowv = sumsq; p.colv = sumsq; p.opart = w.opart; p.spart = w.spart; p.scal_part = w.scal_part;
    SC_TRY(pair_launch<PM_UNIF>(p, st));
    hipLaunchKernelGGL(sum_splits_kernel, dim3((unsigned)sc_cdiv(b * e / 4, 256)), dim3(256), 0, st, w.opart, p.nsplit, b * e, 1.f, wx);
    hipLaunchKernelGGL(sum_splits_kernel, dim3((unsigned)sc_cdiv(b / 4, 256)), dim3(256), 0, st, w.spart, p.nsplit, b, 1.f, rowsum);
    SC_CHECK_LAUNCH();
    return SC_OK;
}

// sparsify_loss: D = X X^T - (2I - 1): loss_out[0] = mean D^2, dx = grad_scale * 4 / B^2 * D X (dx may be null)
int sc_pair_sparsify(const float* x, int64_t b, int64_t e, float grad_scale, float* loss_out, float* dx, float* scratch_be, void* ws, hipStream_t st) {
    PairWs w;
    pair_carve(ws, b, e, w);
    split_launch(x, b, e, w.xh, w.xl, w.xth, w.xtl, st);
    PairParams p = {};
    p.xh = w.xh; p.xl = w.xl; p.yh = w.xh; p.yl = w.xl; p.zth = w.xth; p.ztl = w.xtl; p.Bi = p.Bj = (int)b; p.E = (int)e;
    p.nsplit = pair_nsplit(b); p.jt_per_split = (int)sc_cdiv(b / PT, p.nsplit);
    p.opart = w.opart; p.scal_part = w.scal_part;
    SC_TRY(pair_launch<PM_SPARS>(p, st));
    const float inv = 1.0f / ((float)b * (float)b);
    hipLaunchKernelGGL(sum_scalar_kernel, dim3(1), dim3(256), 0, st, w.scal_part, (int)(b / PT) * p.nsplit, inv, loss_out);
    hipLaunchKernelGGL(sum_splits_kernel, dim3((unsigned)sc_cdiv(b * e / 4, 256)), dim3(256), 0, st, w.opart, p.nsplit, b * e, dx ? grad_scale * 4.0f * inv : 0.f,
                       dx ? dx : scratch_be);
    SC_CHECK_LAUNCH();
    return SC_OK;
}

// ------------------------------------------------------------------------------------------------ row-block forms (sharded loss head)
// Rank r of a data-parallel job owns rows [row0, row0 + bm) of the gathered batch: the sweeps below visit the bm x B rectangle of its
// rows against every column instead of the B x B square.  Same kernels (Bi = bm, Bj = B, diagonal at i + row0 == j), same
// workspace layout as the square forms (the partial buffers of a rectangle never exceed those of the square).
static int pair_nsplit_rows(int64_t bm, int64_t b) {
    const int64_t it = bm / PT, jt = b / PT;
    static const int wgs = [] { const char* e = sc_debug_env("SC_PAIR_WGS"); return e && atoi(e) > 0 ? atoi(e) : 512; }();
    int64_t s = wgs / it;
    if (s < 1) s = 1;
    if (s > jt) s = jt;
    return (int)s;
}
bool sc_pair_rows_supported(int64_t b, int64_t e, int64_t row0, int64_t bm) {
    return sc_pair_supported(b, e) && bm >= PT && bm % PT == 0 && row0 >= 0 && row0 % PT == 0 && row0 + bm <= b;
}

// row LSE of I_R T^T / temp (r of my rows), row LSE of T_R I^T / temp (= column LSE c of my columns), diagonal logits of my rows
int sc_pair_contrastive_rows_stats(const float* img, const float* txt, int64_t b, int64_t e, int64_t row0, int64_t bm, float inv_temp, float* r_rows,
                                   float* c_rows, float* diag_rows, void* ws, hipStream_t st) {
    PairWs w;
    pair_carve(ws, b, e, w);
    split_launch(img, b, e, w.xh, w.xl, nullptr, nullptr, st);
    split_launch(txt, b, e, w.yh, w.yl, nullptr, nullptr, st);
    PairParams p = {};
    p.Bi = (int)bm; p.Bj = (int)b; p.E = (int)e; p.inv_temp = inv_temp; p.diag_off = (int)row0;
    p.nsplit = pair_nsplit_rows(bm, b); p.jt_per_split = (int)sc_cdiv(b / PT, p.nsplit);
    p.rp_m = w.rp_m; p.rp_s = w.rp_s; p.cp_m = w.cp_m; p.cp_s = w.cp_s;
    const int parts = (int)(b / PT);
    const unsigned fb = (unsigned)sc_cdiv(bm, LSE_OUT);
    p.xh = w.xh + row0 * e; p.xl = w.xl + row0 * e; p.yh = w.yh; p.yl = w.yl; p.diag = diag_rows;          // rows = my images, columns = all texts
    const bool s512 = pair_stats512(p);
    const dim3 grid((unsigned)(bm / PT), (unsigned)p.nsplit);
    if (s512) hipLaunchKernelGGL(pair_stats512_kernel, grid, dim3(512), 0, st, p);
    else SC_TRY(pair_launch<PM_CON_STATS>(p, st));
    hipLaunchKernelGGL(lse_final_kernel, dim3(fb), dim3(256), 0, st, w.rp_m, w.rp_s, s512 ? p.nsplit : parts, (int)bm, r_rows);
    p.xh = w.yh + row0 * e; p.xl = w.yl + row0 * e; p.yh = w.xh; p.yl = w.xl; p.diag = nullptr;            // rows = my texts, columns = all images
    if (s512) hipLaunchKernelGGL(pair_stats512_kernel, grid, dim3(512), 0, st, p);
    else SC_TRY(pair_launch<PM_CON_STATS>(p, st));
    hipLaunchKernelGGL(lse_final_kernel, dim3(fb), dim3(256), 0, st, w.rp_m, w.rp_s, s512 ? p.nsplit : parts, (int)bm, c_rows);
    SC_CHECK_LAUNCH();
    return SC_OK;
}

// my rows of dI = G T / temp and of dT = G^T I / temp from the gathered statistics r_all, c_all ([B] each); gv_out (device scalar
// or null) = sum over my image rows and all columns of G v (this rank's part of the temperature gradient)
int sc_pair_contrastive_rows_grad(const float* img, const float* txt, int64_t b, int64_t e, int64_t row0, int64_t bm, float inv_temp, float grad_scale,
                                  const float* r_all, const float* c_all, float* d_img_rows, float* d_txt_rows, float* gv_out, void* ws, hipStream_t st) {
    PairWs w;
    pair_carve(ws, b, e, w);
    split_launch(img, b, e, w.xh, w.xl, w.xth, w.xtl, st);
    split_launch(txt, b, e, w.yh, w.yl, w.yth, w.ytl, st);
    PairParams p = {};
    p.Bi = (int)bm; p.Bj = (int)b; p.E = (int)e; p.inv_temp = inv_temp; p.diag_off = (int)row0;
    p.nsplit = pair_nsplit_rows(bm, b); p.jt_per_split = (int)sc_cdiv(b / PT, p.nsplit);
    p.coef_row = p.coef_col = grad_scale / (2.f * (float)b); p.coef_diag = grad_scale / (float)b;
    p.opart = w.opart; p.scal_part = w.scal_part;
    const int64_t n = bm * e;
    const unsigned rb = (unsigned)sc_cdiv(n / 4, 256);
    p.xh = w.xh + row0 * e; p.xl = w.xl + row0 * e; p.yh = w.yh; p.yl = w.yl; p.zth = w.yth; p.ztl = w.ytl; p.rowv = r_all + row0; p.colv = c_all;
    SC_TRY(pair_launch<PM_CON_GRAD>(p, st));
    hipLaunchKernelGGL(sum_splits_kernel, dim3(rb), dim3(256), 0, st, w.opart, p.nsplit, n, inv_temp, d_img_rows);
    if (gv_out) hipLaunchKernelGGL(sum_scalar_kernel, dim3(1), dim3(256), 0, st, w.scal_part, (int)(bm / PT) * p.nsplit, 1.f, gv_out);
    p.xh = w.yh + row0 * e; p.xl = w.yl + row0 * e; p.yh = w.xh; p.yl = w.xl; p.zth = w.xth; p.ztl = w.xtl; p.rowv = c_all + row0; p.colv = r_all;
    SC_TRY(pair_launch<PM_CON_GRAD>(p, st));
    hipLaunchKernelGGL(sum_splits_kernel, dim3(rb), dim3(256), 0, st, w.opart, p.nsplit, n, inv_temp, d_txt_rows);
    SC_CHECK_LAUNCH();
    return SC_OK;
}

// lunif: my rows of the row sums of W and of W X
int sc_pair_lunif_rows(const float* x, const float* sumsq, int64_t b, int64_t e, int64_t row0, int64_t bm, float t, float* rowsum_rows, float* wx_rows,
                       void* ws, hipStream_t st) {
    PairWs w;
    pair_carve(ws, b, e, w);
    split_launch(x, b, e, w.xh, w.xl, w.xth, w.xtl, st);
    PairParams p = {};
    p.Bi = (int)bm; p.Bj = (int)b; p.E = (int)e; p.t = t; p.diag_off = (int)row0;
    p.nsplit = pair_nsplit_rows(bm, b); p.jt_per_split = (int)sc_cdiv(b / PT, p.nsplit);
    p.xh = w.xh + row0 * e; p.xl = w.xl + row0 * e; p.yh = w.xh; p.yl = w.xl; p.zth = w.xth; p.ztl = w.xtl;
    p.rowv = sumsq + row0; p.colv = sumsq; p.opart = w.opart; p.spart = w.spart; p.scal_part = w.scal_part;
    SC_TRY(pair_launch<PM_UNIF>(p, st));
    hipLaunchKernelGGL(sum_splits_kernel, dim3((unsigned)sc_cdiv(bm * e / 4, 256)), dim3(256), 0, st, w.opart, p.nsplit, bm * e, 1.f, wx_rows);
    hipLaunchKernelGGL(sum_splits_kernel, dim3((unsigned)sc_cdiv(bm / 4, 256)), dim3(256), 0, st, w.spart, p.nsplit, bm, 1.f, rowsum_rows);
    SC_CHECK_LAUNCH();
    return SC_OK;
}
