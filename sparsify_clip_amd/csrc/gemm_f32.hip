// fp32 GEMM on v_mfma_f32_32x32x2_f32 (exact f32, k-ordered fmaf chain) for the fp32 parity path
// and the O(B^2) loss head.  C[M,N] = op(A) op(B), any operand orientation, fused epilogue.
//
// Tile 128x128x32, 256 threads = 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles of 32x32.
// Operands are staged global -> registers -> LDS in a k-major image S[k][m] so that the MFMA
// operand read (one float per lane: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]) is a conflict-free
// ds_read_b32 whatever the storage orientation was.  Register prefetch of tile t+1 overlaps the
// MFMAs of tile t (single LDS buffer, two barriers per k-tile).
#include "common.h"
#include "gemm_epilogue.h"

extern "C" int sc_colsum(const void* x, int dtype, int64_t rows, int64_t n, int64_t ld, float* out, int accumulate, void* ws, size_t ws_bytes,
                         void* stream);

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LD_KC = 129;  // operand stored k-contiguous: transposing ds_write_b32, 129 = 1 (mod 32) -> conflict-free
constexpr int LD_MC = 132;  // operand stored m/n-contiguous: ds_write_b128 needs 16-B aligned rows

struct GemmF32Params {
    const float* A;
    const float* B;
    float* C;
    int M, N, K;
    int64_t lda, ldb, ldc;
    int vecA, vecB;
    EpiParams epi;
};

__device__ __forceinline__ f32x4 ldg4(const float* base, int64_t row_off, int col, int col_limit, bool row_ok, bool vec) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row_ok) {
        const float* p = base + row_off + col;
        if (vec && col + 3 < col_limit) {
            v = *(const f32x4*)p;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (col + j < col_limit) v[j] = p[j];
        }
    }
    return v;
}

// KC = operand stored with k contiguous ([rows][k]); otherwise stored [k][rows].
template <bool KC>
__device__ __forceinline__ void stage_load(f32x4 (&reg)[4], const float* base, int64_t ld, int row0, int row_limit,
                                           int k0, int K, bool vec, int t) {
    if (KC) {
        const int c = t & 7, r = t >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + r + 32 * i;
            reg[i] = ldg4(base, (int64_t)row * ld, k0 + 4 * c, K, row < row_limit, vec);
        }
    } else {
        const int c = t & 31, kr = t >> 5;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + kr + 8 * i;
            reg[i] = ldg4(base, (int64_t)k * ld, row0 + 4 * c, row_limit, k < K, vec);
        }
    }
}

template <bool KC>
__device__ __forceinline__ void stage_store(const f32x4 (&reg)[4], float* S, int t) {
    if (KC) {
        const int c = t & 7, r = t >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) S[(4 * c + j) * LD_KC + r + 32 * i] = reg[i][j];
    } else {
        const int c = t & 31, kr = t >> 5;
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)&S[(kr + 8 * i) * LD_MC + 4 * c] = reg[i];
    }
}

template <bool AK, bool BKC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32Params p) {
    constexpr int LDA_S = AK ? LD_KC : LD_MC;
    constexpr int LDB_S = BKC ? LD_KC : LD_MC;
    __shared__ __attribute__((aligned(16))) float As[BK * LDA_S];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDB_S];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[4], rb[4];
    const int nk = (p.K + BK - 1) / BK;
    stage_load<AK>(ra, p.A, p.lda, m0, p.M, 0, p.K, p.vecA, t);
    stage_load<BKC>(rb, p.B, p.ldb, n0, p.N, 0, p.K, p.vecB, t);

    for (int kt = 0; kt < nk; ++kt) {
        stage_store<AK>(ra, As, t);
        stage_store<BKC>(rb, Bs, t);
        __syncthreads();
        if (kt + 1 < nk) {
            stage_load<AK>(ra, p.A, p.lda, m0, p.M, (kt + 1) * BK, p.K, p.vecA, t);
            stage_load<BKC>(rb, p.B, p.ldb, n0, p.N, (kt + 1) * BK, p.K, p.vecB, t);
        }
        const int am = wm * 64 + (lane & 31), bn = wn * 64 + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int k = 2 * kk + (lane >> 5);
            const float a0 = As[k * LDA_S + am], a1 = As[k * LDA_S + am + 32];
            const float b0 = Bs[k * LDB_S + bn], b1 = Bs[k * LDB_S + bn + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }

    // C/D map of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M && n < p.N) {
                    float* cp = p.C + (int64_t)m * p.ldc + n;
                    *cp = epi_scalar<float>(p.epi, acc[i][j][r], m, n, cp);
                }
            }
        }
}

}  // namespace

int sc_gemm_f32_launch(int trans_a, int trans_b, int64_t m, int64_t n, int64_t k, const float* a, int64_t lda,
                       const float* b, int64_t ldb, float* c, int64_t ldc, const EpiParams& epi, hipStream_t stream) {
    SC_REQUIRE(m > 0 && n > 0 && k > 0, SC_ERR_SHAPE, "sc_gemm_f32: empty problem %lld x %lld x %lld", (long long)m, (long long)n, (long long)k);
    SC_REQUIRE(m < (1ll << 31) && n < (1ll << 31) && k < (1ll << 31), SC_ERR_SHAPE, "sc_gemm_f32: dimension too large");
    SC_REQUIRE(a && b && c, SC_ERR_ARG, "sc_gemm_f32: null operand");
    SC_REQUIRE(lda >= (trans_a ? m : k) && ldb >= (trans_b ? k : n) && ldc >= n, SC_ERR_SHAPE, "sc_gemm_f32: leading dimension too small");
    GemmF32Params p;
    p.A = a; p.B = b; p.C = c;
    p.M = (int)m; p.N = (int)n; p.K = (int)k;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.vecA = (lda % 4 == 0) && sc_aligned(a, 16);
    p.vecB = (ldb % 4 == 0) && sc_aligned(b, 16);
    p.epi = epi;
    dim3 grid((unsigned)sc_cdiv(n, BN), (unsigned)sc_cdiv(m, BM));
    SC_REQUIRE(grid.y <= 65535u, SC_ERR_SHAPE, "sc_gemm_f32: M too large for the grid");
    const bool ak = !trans_a, bk = trans_b != 0;
    if (ak && bk) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, stream, p);
    else if (ak && !bk) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), 0, stream, p);
    else if (!ak && bk) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, stream, p);
    SC_CHECK_LAUNCH();
    return SC_OK;
}

extern "C" int sc_gemm_f32(int trans_a, int trans_b, int64_t m, int64_t n, int64_t k, const float* a, int64_t lda,
                           const float* b, int64_t ldb, float* c, int64_t ldc, const sc_gemm_epilogue* e, void* stream) {
    EpiParams epi;
    SC_TRY(epi_from_abi(e, SC_F32, epi));
    SC_TRY(sc_gemm_f32_launch(trans_a, trans_b, m, n, k, a, lda, b, ldb, c, ldc, epi, (hipStream_t)stream));
    if (epi.colsum)   // the fp32 kernel has no fused column sums: a pass over the stored C
        return sc_colsum(c, SC_F32, m, n, ldc, epi.colsum, epi.colsum_accumulate, epi.colsum_ws, epi.colsum_ws_bytes, stream);
    return SC_OK;
}
