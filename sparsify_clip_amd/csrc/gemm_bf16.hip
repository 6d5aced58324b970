// bf16 MFMA GEMMs for the encoder towers (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
//
//  NT  C[M,N] = A[M,K] B[N,K]^T   both operands K-contiguous: forward linears (x W^T) and, with the [in,out]
//      weight copies, the activation gradients (dX = dY W).
//  TN  C[M,N] = sum_r A[r][m] B[r][n]   both operands contraction-strided: weight gradients dW = dY^T X.
//      The strided operands are staged row-major (coalesced) and read back transposed with
//      ds_read_b64_tr_b16, so no transposed activation copy ever goes through HBM.
//
// Common structure: 128x128 output tile, 256 threads = 4 waves (2x2), each wave a 64x64 sub-tile =
// 4x4 MFMA tiles; K-step 64; two LDS buffers of (16 KiB A + 16 KiB B) filled by global_load_lds_dwordx4
// (LDS image is lane-linear, so the bank swizzle is applied to the per-lane SOURCE address and undone
// on the read - cdna_hip_programming.md rule 21).  The MFMA is issued with the operands swapped
// (D = B_tile A_tile^T) so that each lane's 4 accumulator registers are 4 CONSECUTIVE columns n of one row m:
// the epilogue then loads/stores 8-byte (bf16) or 16-byte (fp32) pieces instead of 2-byte scalars.
// Workgroup ids are remapped so that each XCD walks a contiguous range of tiles (A-panel reuse in its own L2).
#include "common.h"
#include "gemm_epilogue.h"
#include <stdlib.h>

extern "C" int sc_colsum(const void* x, int dtype, int64_t rows, int64_t n, int64_t ld, float* out, int accumulate, void* ws, size_t ws_bytes,
                         void* stream);
int sc_colsum_reduce(const float* partial, int nslab, int64_t n, float* out, int accumulate, hipStream_t st);

namespace {

constexpr int TILE = 128, KSTEP = 64;
constexpr int OPER_BYTES = 16384, BUF_BYTES = 32768;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) bf16x4* ltr_t;

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

struct GemmBf16Params {
    const bf16_t* A;
    const bf16_t* B;
    void* C;
    int M, N, K;          // TN: K is the contraction length R
    int64_t lda, ldb, ldc;
    int tiles_m, tiles_n;
    int splits, k_per_split;   // TN only (k_per_split in units of KSTEP tiles)
    float* partial;            // TN split-R partial slabs [splits][M][N] or null
    float* cs_out;             // TN: optional column sums of A over r, cs_out[m] = cs_beta * cs_out[m] + sum_r A[r][m]  (bias gradient)
    float* cs_partial;         // TN: [splits][M] partial column sums when the contraction is split
    float cs_beta;
    int group_m, group_n;      // NT 256x256: tile-walk cell (row tiles x column tiles an XCD's workgroups cover at a time)
    int half_tiles, half_m0;      // persistent NT kernel: 128x256 tiles that follow the 256x256 ones, covering rows half_m0 .. M - 1
    unsigned long long* stamps;   // diagnostic builds of the persistent NT kernel only (sc_gemm_bf16_nt_stamps): [tile][4] s_memtime values
    // implicit 3x3 convolution (256x128 kernel only; conv_w == 0: a plain GEMM).  A is an NHWC activation with a one-pixel zero border,
    // [batch][conv_h + 2][conv_w + 2][cin]; output row m = (b, y, x) reads, for K-tile kt = tap * kpt + kin, the 64 channels kin of
    // pixel (b, y + tap / 3, x + tap % 3) of the bordered image - no patch matrix exists
    int conv_w, conv_h, conv_kpt_log2;
    EpiParams epi;
};

// bijective XCD-aware remap: blocks b and b+8 share an XCD; give each XCD a contiguous chunk of tiles
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
    const int xcd = id & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
}

// ------------------------------------------------------------------------------------------------ NT
template <bool OUT_F32>
__global__ __launch_bounds__(256, 2) void gemm_bf16_nt_kernel(GemmBf16Params p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF_BYTES];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int m0 = (tile / p.tiles_n) * TILE, n0 = (tile % p.tiles_n) * TILE;

    // staging: one wave instruction = 8 rows x 128 B; lane -> (row l>>3, slot l&7) holds logical chunk slot^(row&7)
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    const bf16_t* ga[4];
    const bf16_t* gb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = wave * 32 + q * 8 + srow;
        ga[q] = p.A + (int64_t)min(m0 + r, p.M - 1) * p.lda + schunk * 8;
        gb[q] = p.B + (int64_t)min(n0 + r, p.N - 1) * p.ldb + schunk * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* abase = smem + buf * BUF_BYTES + (wave * 32) * 128;
        char* bbase = abase + OPER_BYTES;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            glds16(ga[q] + kt * KSTEP, abase + q * 8 * 128);
            glds16(gb[q] + kt * KSTEP, bbase + q * 8 * 128);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / KSTEP;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int frow = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const char* abase = smem + cur * BUF_BYTES + (wm * 64 + frow) * 128;
        const char* bbase = smem + cur * BUF_BYTES + OPER_BYTES + (wn * 64 + frow) * 128;
        // all fragments of this K-tile into registers first: the LDS-DMA of the next tile is then issued with no LDS read
        // behind it, so hipcc places no vmcnt(0) in front of the MFMAs and the loads fly under the 32 MFMAs
        bf16x8 af[2][4], bfr[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int pos = ((4 * s + fq) ^ (lane & 7)) * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) af[s][i] = *(const bf16x8*)(abase + i * 16 * 128 + pos);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[s][j] = *(const bf16x8*)(bbase + j * 16 * 128 + pos);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[s][j], af[s][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // D[n][m]: lane holds m = ..+(lane&15), n = ..+4*(lane>>4)+reg
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + 16 * i + frow;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + 16 * j + 4 * fq;
            if (n >= p.N) continue;
            if (OUT_F32) {
                float* cp = (float*)p.C + (int64_t)m * p.ldc + n;
                *(f32x4*)cp = epi_vec4<bf16_t>(p.epi, acc[i][j], m, n, cp);
            } else {
                bf16_t* cp = (bf16_t*)p.C + (int64_t)m * p.ldc + n;
                io<bf16_t>::st4(cp, epi_vec4<bf16_t>(p.epi, acc[i][j], m, n, nullptr));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ NT, 256x128, 3-stage
// 8 waves as 4(M) x 2(N), each wave a 64x64 sub-tile (4x4 MFMA tiles).  Three LDS stages of (32 KiB A + 16 KiB B):
// the LDS-DMA of K-tile t+2 is issued right after the barrier that publishes tile t, and only a COUNTED vmcnt
// (the youngest stage, 6 DMA per wave, stays in flight) is waited for - HBM/L2 latency hides under two tiles of MFMAs.
// The K loop is unrolled by the stage count so every LDS offset is a compile-time constant.
// Tiles are walked band by band (GROUP_M row tiles x all N) inside each XCD's chunk, GROUP_M x GROUP_N cells at a time, so
// that the ~32 workgroups an XCD runs at once share 8 A panels and 4 B panels and every A panel is fetched once per XCD.
constexpr int T_M = 256, T_N = 128, STAGES = 3, GROUP_N = 4, GROUP_M = 8;
constexpr int STAGE_BYTES = (T_M + T_N) * 128;   // 49152
constexpr int A_BYTES = T_M * 128;

template <bool OUT_F32>
__global__ __launch_bounds__(512, 2) void gemm_bf16_nt256_kernel(GemmBf16Params p) {
    __shared__ __attribute__((aligned(16))) char smem[STAGES * STAGE_BYTES + 8 * 768];   // + per-wave landing pads of the epilogue prefetch
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // XCD chunk -> cell order: GROUP_M row tiles x all column tiles form a band (its A panels stay in the XCD's L2 for the
    // whole sweep over N); inside a band the ~32 resident workgroups cover one GROUP_M x GROUP_N cell at a time.
    const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int band = tile / (GROUP_M * p.tiles_n), r_band = tile - band * (GROUP_M * p.tiles_n);
    const int rows = min(GROUP_M, p.tiles_m - band * GROUP_M);
    const int cell = r_band / (rows * GROUP_N), r_cell = r_band - cell * (rows * GROUP_N);
    const int gw = min(GROUP_N, p.tiles_n - cell * GROUP_N);
    const int m0 = (band * GROUP_M + r_cell / gw) * T_M, n0 = (cell * GROUP_N + r_cell % gw) * T_N;

    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    const bf16_t* ga[4];
    const bf16_t* gb[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int64_t row = min(m0 + wave * 32 + q * 8 + srow, p.M - 1);
        if (p.conv_w) {   // compact output row (b, y, x) -> row of the centre tap's top-left neighbour (b, y, x) in the bordered image
            const int xx = (int)(row % p.conv_w), tt = (int)(row / p.conv_w);
            const int yy = tt % p.conv_h, bb = tt / p.conv_h;
            row = ((int64_t)bb * (p.conv_h + 2) + yy) * (p.conv_w + 2) + xx;
        }
        ga[q] = p.A + row * p.lda + schunk * 8;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) gb[q] = p.B + (int64_t)min(n0 + wave * 16 + q * 8 + srow, p.N - 1) * p.ldb + schunk * 8;
    // element offset of K-tile kt inside an A row (plain GEMM) or to the tap's pixel and channel group (implicit convolution)
    auto a_koff = [&](int kt) -> int64_t {
        if (!p.conv_w) return (int64_t)kt * KSTEP;
        const int tap = kt >> p.conv_kpt_log2, kin = kt & ((1 << p.conv_kpt_log2) - 1);
        const int ty = (tap * 11) >> 5, tx = tap - 3 * ty;
        return ((int64_t)ty * (p.conv_w + 2) + tx) * p.lda + kin * KSTEP;
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / KSTEP;
    const int frow = lane & 15, fq = lane >> 4;
    const int pos0 = ((fq) ^ (lane & 7)) * 16, pos1 = ((4 + fq) ^ (lane & 7)) * 16;

#define NT256_STAGE(S, KT)                                                                            \
    do {                                                                                              \
        char* ab__ = smem + (S) * STAGE_BYTES + (wave * 32) * 128;                                    \
        char* bb__ = smem + (S) * STAGE_BYTES + A_BYTES + (wave * 16) * 128;                          \
        const int64_t ko__ = a_koff(KT);                                                              \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) glds16(ga[q] + ko__, ab__ + q * 8 * 128);       \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) glds16(gb[q] + (KT) * KSTEP, bb__ + q * 8 * 128); \
    } while (0)

#define NT256_READ(S, POS, AF, BF)                                                                    \
    do {                                                                                              \
        const char* ab__ = smem + (S) * STAGE_BYTES + (wm * 64 + frow) * 128 + (POS);                 \
        const char* bb__ = smem + (S) * STAGE_BYTES + A_BYTES + (wn * 64 + frow) * 128 + (POS);       \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) AF[i] = *(const bf16x8*)(ab__ + i * 16 * 128);  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) BF[i] = *(const bf16x8*)(bb__ + i * 16 * 128);  \
    } while (0)
#define NT256_MMA(AF, BF)                                                                             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                     \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                 \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[j], AF[i], acc[i][j], 0, 0, 0)
#define NT256_WEAVE()  /* 2 MFMA : 1 LDS read, eight times */                                         \
    _Pragma("unroll") for (int g__ = 0; g__ < 8; ++g__) {                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                            \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                            \
    }

    // Software pipeline (one barrier per K-tile, LDS reads always under MFMAs):
    //   A: MFMAs of k-substep 0 of tile t  ||  LDS reads of substep 1 of tile t
    //   B: counted vmcnt (tile t+1 landed) + lgkmcnt(0) + barrier   -> stage t%3 is free, stage (t+1)%3 is published
    //   C: LDS-DMA of tile t+3 into stage t%3 (two tiles stay in flight behind the landed one)
    //   D: MFMAs of substep 1 of tile t    ||  LDS reads of substep 0 of tile t+1
#define NT256_BODY(S, KT)                                                                             \
    do {                                                                                              \
        NT256_READ(S, pos1, a1, b1);                                                                  \
        NT256_MMA(a0, b0);                                                                            \
        NT256_WEAVE();                                                                                \
        if ((KT) + 1 < nk) {                                                                          \
            if ((KT) + 2 < nk) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");            \
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                          \
            __builtin_amdgcn_s_barrier();                                                             \
            if ((KT) + 3 < nk) NT256_STAGE(S, (KT) + 3);                                              \
            NT256_READ(((S) + 1) % STAGES, pos0, a0, b0);                                             \
            NT256_MMA(a1, b1);                                                                        \
            NT256_WEAVE();                                                                            \
        } else {                                                                                      \
            NT256_MMA(a1, b1);                                                                        \
        }                                                                                             \
    } while (0)

    bf16x8 a0[4], b0[4], a1[4], b1[4];
    NT256_STAGE(0, 0);
    if (nk > 1) NT256_STAGE(1, 1);
    if (nk > 2) NT256_STAGE(2, 2);
    if (nk > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    NT256_READ(0, pos0, a0, b0);
    // Epilogue operands: the bias goes to registers and the lines of the GELU' pre-activation / residual tile are pulled
    // towards L2 by 4-byte LDS-DMA "touches" (no VGPR destination, landing pad in LDS) one K-tile before they are needed.
    const EpiParams& e = p.epi;
    const int erow = lane >> 3, ecol = (lane & 7) * 8;
    const int n = n0 + wn * 64 + ecol;
    const bool ncol_ok = n < p.N;
    const int nn = ncol_ok ? n : 0;
    f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = {0.f, 0.f, 0.f, 0.f};
    const bool has_pre = e.dgelu_pre != nullptr, has_res = e.resid != nullptr, res_f32 = e.resid_dtype == SC_F32;
    const int last = nk - 1;
#define NT256_PREFETCH()                                                                                             \
    do {                                                                                                             \
        if (e.bias) { bias0 = *(const f32x4*)(e.bias + nn); bias1 = *(const f32x4*)(e.bias + nn + 4); }               \
        char* pad__ = smem + STAGES * STAGE_BYTES + wave * 768;                                                      \
        const int64_t roff__ = (int64_t)min(m0 + wm * 64 + lane, p.M - 1) * e.ld_aux + min(n0 + wn * 64, p.N - 8);    \
        if (has_pre) __builtin_amdgcn_global_load_lds((gptr_t)((const bf16_t*)e.dgelu_pre + roff__), (lptr_t)pad__, 4, 0, 0); \
        if (has_res) {                                                                                               \
            const char* r__ = (const char*)e.resid + roff__ * (res_f32 ? 4 : 2);                                     \
            __builtin_amdgcn_global_load_lds((gptr_t)r__, (lptr_t)(pad__ + 256), 4, 0, 0);                           \
            if (res_f32 && n0 + wn * 64 + 32 < p.N) __builtin_amdgcn_global_load_lds((gptr_t)(r__ + 128), (lptr_t)(pad__ + 512), 4, 0, 0); \
        }                                                                                                            \
    } while (0)
    for (int kt = 0; kt < nk; kt += STAGES) {
        if (kt == last) NT256_PREFETCH();
        NT256_BODY(0, kt);
        if (kt + 1 < nk) {
            if (kt + 1 == last) NT256_PREFETCH();
            NT256_BODY(1, kt + 1);
        }
        if (kt + 2 < nk) {
            if (kt + 2 == last) NT256_PREFETCH();
            NT256_BODY(2, kt + 2);
        }
    }
#undef NT256_PREFETCH
#undef NT256_WEAVE
#undef NT256_MMA
#undef NT256_READ
#undef NT256_BODY
#undef NT256_STAGE

    // Epilogue.  Every lane owns 8 consecutive columns of 8 rows (row = 8*it + lane/8) of its wave's 64x64 sub-tile.
    //  1. all global operands of the epilogue (GELU' pre-activation / residual rows, prefetched towards L2 above) are
    //     requested up front, so at most ONE (L2) latency is exposed per tile instead of one HBM latency per row;
    //  2. the fp32 accumulators cross over to the row-major ownership through a per-wave LDS slab (row stride 68 floats);
    //  3. bias / GELU / GELU' / residual are applied on 16-byte pieces and C is stored 16 bytes per lane, 128 B per row.
    uint4 hpre[8];
    f32x4 res0[8], res1[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int m = min(m0 + wm * 64 + it * 8 + erow, p.M - 1);
        const int64_t off = (int64_t)m * e.ld_aux + nn;
        if (has_pre) hpre[it] = *(const uint4*)((const bf16_t*)e.dgelu_pre + off);
        if (has_res) {
            if (res_f32) {
                res0[it] = *(const f32x4*)((const float*)e.resid + off);
                res1[it] = *(const f32x4*)((const float*)e.resid + off + 4);
            } else {
                res0[it] = io<bf16_t>::ld4((const bf16_t*)e.resid + off);
                res1[it] = io<bf16_t>::ld4((const bf16_t*)e.resid + off + 4);
            }
        }
    }
    __syncthreads();   // every wave is done reading the last K-tile
    float* slab = (float*)smem + wave * (64 * 68);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) *(f32x4*)(slab + (16 * i + frow) * 68 + 16 * j + 4 * fq) = acc[i][j];
    if (ncol_ok) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 8 + erow;
            const int m = m0 + wm * 64 + row;
            if (m >= p.M) continue;
            f32x4 v0 = *(const f32x4*)(slab + row * 68 + ecol) * e.alpha + bias0;
            f32x4 v1 = *(const f32x4*)(slab + row * 68 + ecol + 4) * e.alpha + bias1;
            const int64_t off = (int64_t)m * e.ld_aux + n;
            if (e.pre_out) {
                io<bf16_t>::st4((bf16_t*)e.pre_out + off, v0);
                io<bf16_t>::st4((bf16_t*)e.pre_out + off + 4, v1);
            }
            if (e.act == 1) {
                v0 = gelu_fast4(v0); v1 = gelu_fast4(v1);
            }
            if (has_pre) {
                const uint4 h = hpre[it];
                v0 = gelu_grad_mul4(v0, h.x, h.y); v1 = gelu_grad_mul4(v1, h.z, h.w);
            }
            if (has_res) { v0 += res0[it]; v1 += res1[it]; }
            if (OUT_F32) {
                float* cp = (float*)p.C + (int64_t)m * p.ldc + n;
                if (e.beta != 0.f) { v0 += *(const f32x4*)cp * e.beta; v1 += *(const f32x4*)(cp + 4) * e.beta; }
                *(f32x4*)cp = v0;
                *(f32x4*)(cp + 4) = v1;
            } else {
                uint4 u;
                u.x = pack2_bf16(v0[0], v0[1]);
                u.y = pack2_bf16(v0[2], v0[3]);
                u.z = pack2_bf16(v1[0], v1[1]);
                u.w = pack2_bf16(v1[2], v1[3]);
                *(uint4*)((bf16_t*)p.C + (int64_t)m * p.ldc + n) = u;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ NT, 256x256, 2-stage
// Twice the B panel per workgroup: 128 flop per LDS-DMA byte instead of 85.  tools/ingest_bench.hip takes the main loop
// apart on the fc1 shape: with the 256x128 tile, LDS-DMA + fragment reads + MFMAs (no epilogue) run at 1.08 PFLOP/s-equivalent,
// with this tile and wave layout at 1.33 - the loop is bound by how many operand bytes one CU can keep in flight.
// 8 waves as 2(M) x 4(N), each wave a 128x64 sub-tile = 8x4 MFMA tiles.  Two LDS stages of (32 KiB A + 32 KiB B), K-step 64,
// one barrier per K-tile placed between its two 32-MFMA sub-steps.
// The 128 accumulator registers of a wave are the physical AGPRs a0..a127, named directly in inline assembly: MFMA tile (i, j)
// lives in a[4(4i+j) : 4(4i+j)+3].  Left to itself hipcc keeps the accumulators of such a loop in VGPRs, runs out of them and
// shuttles values through v_accvgpr_read/write and scratch around every MFMA.  MFMAs and fragment reads are therefore emitted
// as inline assembly in their final order: row tile i's four MFMAs, then the read that reloads a[i] for the NEXT sub-step
// into the same registers (dead by then), the B fragments ping-pong between two sets - 64 fragment VGPRs in all.
// The compiler never sees the AGPRs, so the kernel must not spill (checked at build time) and the hazards it can no longer
// see are handled by hand: s_waitcnt lgkmcnt before the first MFMA that consumes a fragment, s_nop before the read-back.
constexpr int B_M = 256, B_N = 256, B_STAGE = (B_M + B_N) * 128, B_A = B_M * 128, B_GROUP_M = 8, B_GROUP_N = 4, B_LDS = 8 * 64 * 68 * 4;

// Clobber list of the accumulator AGPRs: attached to EVERY inline-assembly statement that reads or writes them, so that the
// register allocator (which freely parks long-lived values in AGPRs on gfx90a+) can keep nothing of its own in a0..a127
// across the main loop or across the accumulator read-back.
#define SC_ACC_AGPRS \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", \
    "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", \
    "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", \
    "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", \
    "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", \
    "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", \
    "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", \
    "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"

// ZERO: the tile's first k-step - the accumulator input is the constant 0, so the 128 v_accvgpr_write that cleared the accumulators in front
// of every tile (on the critical path of every tile but a workgroup's first: its K-tiles are resident when the tile starts) are not needed
template <int F, bool ZERO = false>
__device__ __forceinline__ void ntb_mfma(const bf16x8 (&a)[8], const bf16x8 (&b)[4]) {
    if constexpr (ZERO) asm volatile("v_mfma_f32_16x16x32_bf16 a[%2:%3], %0, %1, 0" : : "v"(b[F % 4]), "v"(a[F / 4]), "n"(4 * F), "n"(4 * F + 3) : SC_ACC_AGPRS);
    else asm volatile("v_mfma_f32_16x16x32_bf16 a[%2:%3], %0, %1, a[%2:%3]" : : "v"(b[F % 4]), "v"(a[F / 4]), "n"(4 * F), "n"(4 * F + 3) : SC_ACC_AGPRS);
}
template <int OFF>
__device__ __forceinline__ void ntb_read(bf16x8& dst, unsigned base) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(OFF) : "memory");
}
// one sub-step: 8 groups of 4 MFMAs; when LOAD, group I is followed by the read of next sub-step's a[I] (in place) and, for
// I < 4, of its B fragment I into the other B set.  The 12 reads of a sub-step are therefore issued in the order
// a0 b0 a1 b1 a2 b2 a3 b3 a4 a5 a6 a7.  LATE (measured, no gain, not used): enter the sub-step with a4..a7 still in flight
// (s_waitcnt lgkmcnt(4)); LDS reads return in order, so before group I >= 4 "at most 11 outstanding" proves a[I] has landed.
// NG = row tiles of the wave (8: 128-row wave tile; 4: the 64-row wave tile of the persistent kernel's half tiles)
template <int I, bool LOAD, bool LATE = false, int NG = 8, bool ZERO = false>
__device__ __forceinline__ void ntb_substep(bf16x8 (&a)[8], const bf16x8 (&b)[4], bf16x8 (&bn)[4], unsigned abase, unsigned bbase) {
    if constexpr (LATE && I >= 4) {
        if constexpr (LOAD) asm volatile("s_waitcnt lgkmcnt(11)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(7 - I) : "memory");
    }
    ntb_mfma<4 * I, ZERO>(a, b); ntb_mfma<4 * I + 1, ZERO>(a, b); ntb_mfma<4 * I + 2, ZERO>(a, b); ntb_mfma<4 * I + 3, ZERO>(a, b);
    if constexpr (LOAD) {
        ntb_read<I * 2048>(a[I], abase);
        if constexpr (I < 4) ntb_read<I * 2048>(bn[I], bbase);
    }
    if constexpr (I + 1 < NG) ntb_substep<I + 1, LOAD, LATE, NG, ZERO>(a, b, bn, abase, bbase);
}
template <int N>
__device__ __forceinline__ void ntb_zero() {
    asm volatile("v_accvgpr_write_b32 a[%0], 0\n v_accvgpr_write_b32 a[%1], 0\n v_accvgpr_write_b32 a[%2], 0\n v_accvgpr_write_b32 a[%3], 0"
                 : : "n"(N), "n"(N + 1), "n"(N + 2), "n"(N + 3) : SC_ACC_AGPRS);
    if constexpr (N + 4 < 128) ntb_zero<N + 4>();
}
template <int T>
__device__ __forceinline__ f32x4 ntb_acc() {   // accumulator tile T = 4 i + j
    f32x4 v;
    asm volatile("v_accvgpr_read_b32 %0, a[%4]\n v_accvgpr_read_b32 %1, a[%5]\n v_accvgpr_read_b32 %2, a[%6]\n v_accvgpr_read_b32 %3, a[%7]"
                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]) : "n"(4 * T), "n"(4 * T + 1), "n"(4 * T + 2), "n"(4 * T + 3) : SC_ACC_AGPRS);
    return v;
}
// accumulator tile T -> LDS (16 bytes per lane at byte address `lds`) straight from the AGPRs: DS instructions take AGPR data operands,
// so the 128 v_accvgpr_read of a tile's read-back (and their VGPRs) are not needed.  Ordered against the compiler's own LDS accesses
// by the "memory" clobber; LDS operations of one wave execute in issue order, so the slab reads behind it see the data.
template <int T>
__device__ __forceinline__ void ntb_acc_to_lds(unsigned lds) {
    asm volatile("ds_write_b128 %0, a[%1:%2]" : : "v"(lds), "n"(4 * T), "n"(4 * T + 3) : "memory", SC_ACC_AGPRS);
}
template <int T, int H>
__device__ __forceinline__ void ntb_to_slab(float* slab_lane) {   // rows 64 H .. 64 H + 63 of the wave's sub-tile: i = 4 H + T / 4, j = T % 4
    *(f32x4*)(slab_lane + (16 * (T / 4)) * 68 + 16 * (T % 4)) = ntb_acc<16 * H + T>();
    if constexpr (T + 1 < 16) ntb_to_slab<T + 1, H>(slab_lane);
}

template <bool OUT_F32>
__global__ __launch_bounds__(512, 2) void gemm_bf16_nt_big_kernel(GemmBf16Params p) {
    __shared__ __attribute__((aligned(16))) char smem[B_LDS];   // 139264 >= 2 stages (131072); eight 64x68-float epilogue slabs
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int GM = p.group_m, GN = p.group_n;
    const int band = tile / (GM * p.tiles_n), r_band = tile - band * (GM * p.tiles_n);
    const int rows = min(GM, p.tiles_m - band * GM);
    const int cell = r_band / (rows * GN), r_cell = r_band - cell * (rows * GN);
    const int gw = min(GN, p.tiles_n - cell * GN);
    const int m0 = (band * GM + r_cell / gw) * B_M, n0 = (cell * GN + r_cell % gw) * B_N;

    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    const bf16_t* ga[4];
    const bf16_t* gb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        ga[q] = p.A + (int64_t)min(m0 + wave * 32 + q * 8 + srow, p.M - 1) * p.lda + schunk * 8;
        gb[q] = p.B + (int64_t)min(n0 + wave * 32 + q * 8 + srow, p.N - 1) * p.ldb + schunk * 8;
    }
    // reserves a0..a127 in the kernel descriptor (the compiler allocates what it sees clobbered)
    asm volatile("" ::: SC_ACC_AGPRS);

    const int nk = p.K / KSTEP;
    const int frow = lane & 15, fq = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const unsigned pos0 = ((fq) ^ (lane & 7)) * 16, pos1 = ((4 + fq) ^ (lane & 7)) * 16;
    const unsigned fa = lds0 + (wm * 128 + frow) * 128, fb = lds0 + B_A + (wn * 64 + frow) * 128;
    // fragment i of stage S, sub-step s: base[S][s] + i * 2048 (the DS offset field holds 16 bits, so each stage has its own base)
    const unsigned fa00 = fa + pos0, fa01 = fa + pos1, fa10 = fa + B_STAGE + pos0, fa11 = fa + B_STAGE + pos1;
    const unsigned fb00 = fb + pos0, fb01 = fb + pos1, fb10 = fb + B_STAGE + pos0, fb11 = fb + B_STAGE + pos1;

#define NTB_STAGE(S, KT)                                                                              \
    do {                                                                                              \
        char* ab__ = smem + (S) * B_STAGE + (wave * 32) * 128;                                        \
        char* bb__ = ab__ + B_A;                                                                      \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) glds16(ga[q] + (KT) * KSTEP, ab__ + q * 8 * 128); \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) glds16(gb[q] + (KT) * KSTEP, bb__ + q * 8 * 128); \
    } while (0)
    // Stage S holds tile KT.  Sub-step 0 of tile KT runs while the fragments of its sub-step 1 are read (stage S); the barrier
    // publishes tile KT+1 (vmcnt(0): the only tile in flight) and retires every read of stage S (lgkmcnt(0)), so the LDS-DMA
    // of tile KT+2 may overwrite stage S right behind it; sub-step 1 then reads the first fragments of tile KT+1 from S^1.
#define NTB_FULL(S, KT, A_S1, B_S1, A_N0, B_N0)   /* tile KT is not the last one */                   \
    do {                                                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
        ntb_substep<0, true>(a, b0, b1, A_S1, B_S1);                                                  \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                   \
        __builtin_amdgcn_s_barrier();                                                                 \
        if ((KT) + 2 < nk) NTB_STAGE(S, (KT) + 2);                                                    \
        ntb_substep<0, true>(a, b1, b0, A_N0, B_N0);                                                  \
    } while (0)
#define NTB_LAST(A_S1, B_S1)                                                                          \
    do {                                                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
        ntb_substep<0, true>(a, b0, b1, A_S1, B_S1);                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
        ntb_substep<0, false>(a, b1, b0, 0u, 0u);                                                     \
    } while (0)

    bf16x8 a[8], b0[4], b1[4];
    // both stages are free at the start: K-tile 1 is requested together with K-tile 0 (8 LDS-DMA instructions per wave and stage),
    // the accumulators are cleared under the first latency, and only K-tile 0 is waited for
    NTB_STAGE(0, 0);
    if (nk > 1) NTB_STAGE(1, 1);
    ntb_zero<0>();
    if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // same issue order as inside a sub-step (a0 b0 a1 b1 a2 b2 a3 b3 a4 a5 a6 a7): the loop's counted waits rely on it
    ntb_read<0>(a[0], fa00); ntb_read<0>(b0[0], fb00); ntb_read<2048>(a[1], fa00); ntb_read<2048>(b0[1], fb00);
    ntb_read<4096>(a[2], fa00); ntb_read<4096>(b0[2], fb00); ntb_read<6144>(a[3], fa00); ntb_read<6144>(b0[3], fb00);
    ntb_read<8192>(a[4], fa00); ntb_read<10240>(a[5], fa00); ntb_read<12288>(a[6], fa00); ntb_read<14336>(a[7], fa00);
    // the loop only runs bodies that are followed by another tile, so every exit edge leads to code that starts with a wait:
    // no inline-asm LDS read is in flight when compiler-scheduled code (the epilogue) begins
    int kt = 0;
    for (; kt + 2 < nk; kt += 2) {
        NTB_FULL(0, kt, fa01, fb01, fa10, fb10);
        NTB_FULL(1, kt + 1, fa11, fb11, fa00, fb00);
    }
    const bool two_left = kt + 1 < nk;   // one straight-line tail (no second copy of the last body that a path could bypass)
    if (two_left) NTB_FULL(0, kt, fa01, fb01, fa10, fb10);
    NTB_LAST(two_left ? fa11 : fa01, two_left ? fb11 : fb01);
    asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");   // no LDS read in flight on any exit edge; the last MFMAs retire before the accumulators are read back
#undef NTB_LAST
#undef NTB_FULL
#undef NTB_STAGE

    // Epilogue: as in the 256x128 kernel (per-wave 64x64 LDS slab, row-major 16-byte pieces), two passes of 64 rows per wave.
    const EpiParams& e = p.epi;
    const int erow = lane >> 3, ecol = (lane & 7) * 8;
    const int n = n0 + wn * 64 + ecol;
    const bool ncol_ok = n < p.N;
    const int nn = ncol_ok ? n : 0;
    f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = {0.f, 0.f, 0.f, 0.f};
    const bool has_pre = e.dgelu_pre != nullptr, has_res = e.resid != nullptr, res_f32 = e.resid_dtype == SC_F32;
    if (e.bias) { bias0 = *(const f32x4*)(e.bias + nn); bias1 = *(const f32x4*)(e.bias + nn + 4); }
    __syncthreads();   // every wave is done reading the last K-tile
    float* slab = (float*)smem + wave * (64 * 68);
    // fused column sums of the stored output (bias gradient of the producing layer): per lane 8 columns over all of its rows
    const bool do_cs = e.cs_partial != nullptr;
    f32x4 cs0 = {0.f, 0.f, 0.f, 0.f}, cs1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int mw = m0 + wm * 128 + h * 64;
        if (h == 0) ntb_to_slab<0, 0>(slab + frow * 68 + 4 * fq);
        else ntb_to_slab<0, 1>(slab + frow * 68 + 4 * fq);
        if (ncol_ok) {
#pragma unroll
          for (int g = 0; g < 2; ++g) {   // 4 rows per lane at a time: their GELU' / residual operands are requested together
            uint4 hpre[4];
            f32x4 res0[4], res1[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int m = min(mw + (4 * g + it) * 8 + erow, p.M - 1);
                const int64_t off = (int64_t)m * e.ld_aux + nn;
                if (has_pre) hpre[it] = *(const uint4*)((const bf16_t*)e.dgelu_pre + off);
                if (has_res) {
                    if (res_f32) {
                        res0[it] = *(const f32x4*)((const float*)e.resid + off);
                        res1[it] = *(const f32x4*)((const float*)e.resid + off + 4);
                    } else {
                        res0[it] = io<bf16_t>::ld4((const bf16_t*)e.resid + off);
                        res1[it] = io<bf16_t>::ld4((const bf16_t*)e.resid + off + 4);
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = (4 * g + it) * 8 + erow;
                const int m = mw + row;
                if (m >= p.M) continue;
                f32x4 v0 = *(const f32x4*)(slab + row * 68 + ecol) * e.alpha + bias0;
                f32x4 v1 = *(const f32x4*)(slab + row * 68 + ecol + 4) * e.alpha + bias1;
                const int64_t off = (int64_t)m * e.ld_aux + n;
                if (e.pre_out) {
                    io<bf16_t>::st4((bf16_t*)e.pre_out + off, v0);
                    io<bf16_t>::st4((bf16_t*)e.pre_out + off + 4, v1);
                }
                if (e.act == 1) {
                    v0 = gelu_fast4(v0); v1 = gelu_fast4(v1);
                }
                if (has_pre) {
                    const uint4 hh = hpre[it];
                    v0 = gelu_grad_mul4(v0, hh.x, hh.y); v1 = gelu_grad_mul4(v1, hh.z, hh.w);
                }
                if (has_res) { v0 += res0[it]; v1 += res1[it]; }
                if (OUT_F32) {
                    float* cp = (float*)p.C + (int64_t)m * p.ldc + n;
                    if (e.beta != 0.f) { v0 += *(const f32x4*)cp * e.beta; v1 += *(const f32x4*)(cp + 4) * e.beta; }
                    *(f32x4*)cp = v0;
                    *(f32x4*)(cp + 4) = v1;
                    if (do_cs) { cs0 += v0; cs1 += v1; }
                } else {
                    uint4 u;
                    u.x = pack2_bf16(v0[0], v0[1]);
                    u.y = pack2_bf16(v0[2], v0[3]);
                    u.z = pack2_bf16(v1[0], v1[1]);
                    u.w = pack2_bf16(v1[2], v1[3]);
                    *(uint4*)((bf16_t*)p.C + (int64_t)m * p.ldc + n) = u;
                    if (do_cs) {   // the values as stored (bf16-rounded): identical to a pass over C
                        cs0[0] += __uint_as_float(u.x << 16); cs0[1] += __uint_as_float(u.x & 0xffff0000u);
                        cs0[2] += __uint_as_float(u.y << 16); cs0[3] += __uint_as_float(u.y & 0xffff0000u);
                        cs1[0] += __uint_as_float(u.z << 16); cs1[1] += __uint_as_float(u.z & 0xffff0000u);
                        cs1[2] += __uint_as_float(u.w << 16); cs1[3] += __uint_as_float(u.w & 0xffff0000u);
                    }
                }
            }
          }
        }
    }
    if (do_cs) {   // the 8 row groups of a wave (lane bits 3..5) in a fixed order, then one partial row per (row tile, wave row)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            cs0[j] += __shfl_xor(cs0[j], 8, 64);  cs1[j] += __shfl_xor(cs1[j], 8, 64);
            cs0[j] += __shfl_xor(cs0[j], 16, 64); cs1[j] += __shfl_xor(cs1[j], 16, 64);
            cs0[j] += __shfl_xor(cs0[j], 32, 64); cs1[j] += __shfl_xor(cs1[j], 32, 64);
        }
        if (lane < 8 && ncol_ok) {
            float* dst = e.cs_partial + (int64_t)((m0 / B_M) * 2 + wm) * p.N + n;
            *(f32x4*)dst = cs0;
            *(f32x4*)(dst + 4) = cs1;
        }
    }
}

// LDS-DMA with a uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset, written out so that the compiler cannot widen
// the eight offsets back into eight 64-bit pointers (16 VGPRs the AGPR kernels do not have).  M0 = LDS destination of the wave.
__device__ __forceinline__ void glds16_saddr(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");   // M0 is written; the kernels using it have no compiler-generated M0 user
}

// ------------------------------------------------------------------------------------------------ NT, 256x256, persistent
// The kernel above pays, per 256x256 tile at K = 768, ~12 us of prologue + epilogue with idle MFMA units against ~18 us of main
// loop (workgroup launch, first LDS-DMA round trip, epilogue through slabs that alias the operand stages, teardown).  This
// kernel keeps its main loop (same wave tile, same AGPR accumulators, same hand-placed MFMA / ds_read order) and removes what
// sits between two tiles, for the problems the training step is made of - whole 256x256 tiles and one of four epilogues:
//   * ONE workgroup per CU walks a list of tiles (virtual block ids blockIdx.x + r * gridDim.x through the same XCD-aware
//     band/cell remap, so every tile still runs on the XCD the one-tile-per-workgroup kernel would give it);
//   * the epilogue slabs do not alias the operand stages: LDS = 2 stages (128 KiB) + eight 4-KiB per-wave slabs = 160 KiB.
//     A slab holds ONE 16-row MFMA row tile (16 x 64 fp32, 16-byte pieces XOR-swizzled by the row instead of padded), so the
//     accumulators cross over to the row-major ownership in eight small passes per wave;
//   * hence both stages are free as soon as the last K-tile's fragments are in registers: between the two sub-steps of the
//     last K-tile (one extra barrier) the wave requests K-tiles 0 and 1 of its NEXT tile, which land under the last 32 MFMAs
//     and the whole epilogue - the next main loop starts on resident data;
//   * the epilogue's global operands (GELU' pre-activation, fp32 residual) are requested one row tile ahead;
//   * with whole tiles nothing is clamped: LDS-DMA sources are a per-tile SGPR base + eight per-lane 32-bit offsets computed
//     once per kernel, epilogue addresses a per-tile SGPR base + one per-lane offset.
// The epilogue kind is a template parameter (each instance keeps only its own operands alive).  Everything else - ragged
// shapes, alpha / beta, other epilogue combinations - stays on the one-tile-per-workgroup kernel above.
constexpr int P_SLAB = 4096, P_LDS = 2 * B_STAGE + 8 * P_SLAB;   // 163840 B = all of the CU's LDS
enum { NTP_BIAS = 0, NTP_GELU_PRE = 1, NTP_DGELU = 2, NTP_RESID = 3 };

template <int EPI>
struct NtpAux {   // epilogue operands of one 16-row tile: two groups of 8 rows, 8 columns per lane
    uint4 hpre[EPI == NTP_DGELU ? 2 : 1];
    f32x4 r0[EPI == NTP_RESID ? 2 : 1], r1[EPI == NTP_RESID ? 2 : 1];
};

struct NtpEpi {   // per-tile epilogue context: uniform bases (SGPRs) + this lane's offsets
    char* c;                // C tile origin of this wave's 128x64 sub-tile
    const char* aux;        // dgelu_pre (bf16) or resid (fp32), same origin
    char* pre;              // pre_out (bf16), same origin
    unsigned row_bytes_c, row_bytes_aux, row_bytes_pre;   // bytes per matrix row
    unsigned voff_c, voff_aux, voff_pre;                  // lane: (erow * ld + ecol) * element size
    unsigned slab_wr[4];    // LDS byte addresses of this lane's four 16-byte pieces (MFMA ownership) of a row tile in the slab
    const char* slab;       // wave's slab
    f32x4 bias0, bias1;
};

__device__ __forceinline__ uint4 pack8_bf16(const f32x4& v0, const f32x4& v1) {
    uint4 u;
    u.x = pack2_bf16(v0[0], v0[1]);
    u.y = pack2_bf16(v0[2], v0[3]);
    u.z = pack2_bf16(v1[0], v1[1]);
    u.w = pack2_bf16(v1[2], v1[3]);
    return u;
}

// 16-byte unit of piece p (4 fp32 columns) of slab row `row`.  bf16 outputs: a lane stores 8 consecutive columns (pieces 2 ep, 2 ep + 1 ->
// one 16-byte store, 128 contiguous bytes per row and instruction) and p ^ row keeps both the accumulator writes (8 rows, one piece) and
// the row-major reads apart.  fp32 output (NTP_RESID): with the same ownership every load / store instruction touched HALF of each 32-byte
// run (16 of every 32 bytes: twice the memory requests for the same bytes, and the epilogue was bound by the CU's request rate, not by the
// chip - tools/epi_half_chip.py); there a lane owns pieces ep and 8 + ep, so one instruction covers 128 contiguous bytes of a row, and the
// unit is p ^ (row & 7) ^ 8 * bit 1 of row (conflict-free for the 8-row writes and for the 4-row x 4-piece groups of a ds_read_b128).
template <int EPI>
__device__ __forceinline__ unsigned ntp_slab_unit(unsigned p, unsigned row) {
    if constexpr (EPI == NTP_RESID) return p ^ (row & 7u) ^ (((row >> 1) & 1u) << 3);
    else return (p ^ row) & 15u;
}

template <int EPI, int I>
__device__ __forceinline__ void ntp_aux_load(NtpAux<EPI>& x, const NtpEpi& c) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const char* row = c.aux + (size_t)(16 * I + 8 * g) * c.row_bytes_aux;   // uniform
        if constexpr (EPI == NTP_DGELU) x.hpre[g] = *(const uint4*)(row + c.voff_aux);
        if constexpr (EPI == NTP_RESID) {
            x.r0[g] = *(const f32x4*)(row + c.voff_aux);
            x.r1[g] = *(const f32x4*)(row + c.voff_aux + 128);
        }
    }
}

// row tile I of the wave's 128x64 sub-tile: accumulators -> slab (MFMA ownership: row lane & 15, columns 16 j + 4 (lane >> 4) ..),
// slab -> registers (row-major ownership: row lane >> 3 (+8), 8 columns 8 (lane & 7) ..), epilogue arithmetic, 16-byte stores.
template <int EPI, int I>
__device__ __forceinline__ void ntp_row_tile(const NtpEpi& c, int lane, const NtpAux<EPI>& x, bool do_cs, f32x4& cs0, f32x4& cs1) {
    const int erow = lane >> 3, ep = lane & 7;
    ntb_acc_to_lds<4 * I + 0>(c.slab_wr[0]);
    ntb_acc_to_lds<4 * I + 1>(c.slab_wr[1]);
    ntb_acc_to_lds<4 * I + 2>(c.slab_wr[2]);
    ntb_acc_to_lds<4 * I + 3>(c.slab_wr[3]);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int row = 8 * g + erow;
        const char* rd = c.slab + row * 256;
        const unsigned p0 = EPI == NTP_RESID ? ep : 2 * ep, p1 = EPI == NTP_RESID ? 8 + ep : 2 * ep + 1;
        f32x4 v0 = *(const f32x4*)(rd + (ntp_slab_unit<EPI>(p0, row) << 4)) + c.bias0;
        f32x4 v1 = *(const f32x4*)(rd + (ntp_slab_unit<EPI>(p1, row) << 4)) + c.bias1;
        const size_t rsel = (size_t)(16 * I + 8 * g);
        if constexpr (EPI == NTP_GELU_PRE) {
            *(uint4*)(c.pre + rsel * c.row_bytes_pre + c.voff_pre) = pack8_bf16(v0, v1);
            v0 = gelu_fast4(v0); v1 = gelu_fast4(v1);
        }
        if constexpr (EPI == NTP_DGELU) {
            const uint4 hh = x.hpre[g];
            v0 = gelu_grad_mul4(v0, hh.x, hh.y); v1 = gelu_grad_mul4(v1, hh.z, hh.w);
        }
        if constexpr (EPI == NTP_RESID) {
            v0 += x.r0[g]; v1 += x.r1[g];
            char* cp = c.c + rsel * c.row_bytes_c + c.voff_c;
            *(f32x4*)cp = v0;
            *(f32x4*)(cp + 128) = v1;
        } else {
            const uint4 u = pack8_bf16(v0, v1);
            *(uint4*)(c.c + rsel * c.row_bytes_c + c.voff_c) = u;
            if constexpr (EPI == NTP_DGELU) {
                if (do_cs) {   // the values as stored (bf16-rounded): identical to a pass over C
                    cs0[0] += __uint_as_float(u.x << 16); cs0[1] += __uint_as_float(u.x & 0xffff0000u);
                    cs0[2] += __uint_as_float(u.y << 16); cs0[3] += __uint_as_float(u.y & 0xffff0000u);
                    cs1[0] += __uint_as_float(u.z << 16); cs1[1] += __uint_as_float(u.z & 0xffff0000u);
                    cs1[2] += __uint_as_float(u.w << 16); cs1[3] += __uint_as_float(u.w & 0xffff0000u);
                }
            }
        }
    }
}

// The epilogue's global operands (GELU' pre-activation, fp32 residual) are requested D row tiles ahead, in a ring of D register sets.  The
// vector-memory counter retires loads and stores in issue order, so a wait for the operands of row tile I also waits for every store
// issued in front of their request: with the request of row tile I + D placed behind the stores of row tile I, a store has D row
// tiles' time to reach memory before anything waits for it (D = 1, rounds 1-2: the out_proj / fc2 epilogues ran at the latency of
// one store round trip per row tile, 11 B / clk / CU with the memory system far from saturated - delaying half of the workgroups by
// half a tile period changed nothing, tools/desync_bench.py, round 3).
template <int EPI> constexpr int ntp_aux_depth() { return EPI == NTP_RESID ? 3 : EPI == NTP_DGELU ? 4 : 1; }

template <int EPI, int I, int NP, int D>
__device__ __forceinline__ void ntp_epilogue_rows(const NtpEpi& c, int lane, NtpAux<EPI> (&ring)[D], bool do_cs, f32x4& cs0, f32x4& cs1) {
    ntp_row_tile<EPI, I>(c, lane, ring[I % D], do_cs, cs0, cs1);
    if constexpr (I + D < NP && (EPI == NTP_DGELU || EPI == NTP_RESID)) ntp_aux_load<EPI, I + D>(ring[I % D], c);
    if constexpr (I + 1 < NP) ntp_epilogue_rows<EPI, I + 1, NP, D>(c, lane, ring, do_cs, cs0, cs1);
}
template <int EPI, int I, int NP, int D>
__device__ __forceinline__ void ntp_aux_fill(NtpAux<EPI> (&ring)[D], const NtpEpi& c) {
    if constexpr (I < D && I < NP) {
        ntp_aux_load<EPI, I>(ring[I], c);
        ntp_aux_fill<EPI, I + 1, NP, D>(ring, c);
    }
}

__device__ __forceinline__ unsigned long long ntp_stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

struct NtpTile {   // one entry of a workgroup's tile list (wave-uniform; three ints, so that copies stay in SGPRs)
    int m0, n0;
    int half;        // != 0: 128 x 256 instead of 256 x 256
};
// Ticket of the dynamic tile order (see gemm_bf16_nt_pers_kernel).  request: lane 0 of the calling wave adds 1 to *counter, the old
// value arrives in `ret` about a microsecond later - written out so that nothing waits for it here (the compiler's own atomicAdd is
// followed by s_waitcnt vmcnt(0) at once).  publish: called behind the NEXT K-tile's s_waitcnt vmcnt(0) (which the returned value
// has shared with that K-tile's LDS-DMA), stores it to the LDS word `box`.  Between the two calls `ret` is a register with a
// write in flight: the build-time audit fails if any instruction names it there.
__device__ __forceinline__ void ntp_ticket_request(unsigned* counter, int& ret) {
    unsigned long long saved;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tv_mov_b32 %0, 1\n\tglobal_atomic_add %0, %2, %0, %3 sc0\n\ts_mov_b64 exec, %1"
                 : "=&v"(ret), "=&s"(saved) : "v"(0u), "s"(counter) : "memory");
}
__device__ __forceinline__ void ntp_ticket_publish(unsigned box, int ret) {
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_write_b32 %1, %2\n\ts_mov_b64 exec, %0" : "=&s"(saved) : "v"(box), "v"(ret) : "memory");
}

// Tiles 0 .. full-1 are 256x256 (band / cell walk of the one-tile-per-workgroup kernel), tiles full .. full+half_tiles-1 are
// 128x256 and cover the rows from half_m0 on: the launcher turns the last, partly filled round of 256x256 tiles into (at most
// one round of) half tiles, so that the tail of the launch costs half a tile time per CU instead of a whole one.
__device__ __forceinline__ NtpTile ntp_tile_of(const GemmBf16Params& p, int vb, int full) {
    NtpTile t;
    if (vb < full) {
        const int GM = p.group_m, GN = p.group_n;
        const int tile = xcd_remap(vb, full);
        const int band = tile / (GM * p.tiles_n), rb = tile - band * (GM * p.tiles_n);
        const int rows = min(GM, p.tiles_m - band * GM);
        const int cell = rb / (rows * GN), rc = rb - cell * (rows * GN);
        const int gw = min(GN, p.tiles_n - cell * GN);
        t.m0 = (band * GM + rc / gw) * B_M; t.n0 = (cell * GN + rc % gw) * B_N;
        t.half = 0;
    } else {
        const int h = xcd_remap(vb - full, p.half_tiles);   // consecutive column tiles of one half row panel stay on one XCD
        t.m0 = p.half_m0 + (h / p.tiles_n) * (B_M / 2); t.n0 = (h % p.tiles_n) * B_N;
        t.half = 1;
    }
    return t;
}

// The persistent kernel reads its arguments from the kernarg segment WHEN IT NEEDS THEM instead of holding ~35 of them in SGPRs for
// the whole launch (the kernel sits at the 102-SGPR limit; what does not fit is kept in VGPR lanes and - for the LDS-DMA bases -
// in VGPRs, with v_readfirstlane in front of every use).  ntp_args launders the segment pointer, so loads behind it cannot be
// merged with loads in front of it: called at the start of a tile, in front of the next-tile computation and in front of the
// epilogue, each phase loads its own few arguments (scalar loads that hit the constant cache) and nothing else is live across it.
typedef const __attribute__((address_space(4))) GemmBf16Params* ntp_kargs_t;
__device__ __forceinline__ const GemmBf16Params& ntp_args() {
    ntp_kargs_t k = (ntp_kargs_t)__builtin_amdgcn_kernarg_segment_ptr();   // the parameter struct is the first (only) explicit argument
    asm volatile("" : "+s"(k));
    return *(const GemmBf16Params*)k;
}
// wave-uniform by construction; said explicitly, so that the value is in SGPRs whatever the optimiser made of the tile walk (at
// the SGPR limit it keeps uniform values in VGPRs, which an "s" operand of inline assembly does not survive)
__device__ __forceinline__ const char* ntp_uniform(const char* ptr) {
    const unsigned long long v = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}
// LDS-DMA of K-tile KT of tile `t` into stage S.  Full tile: wave w brings A rows 32 w .. 32 w + 31 (4 instructions) and the same B
// rows; half tile: A rows 16 w .. 16 w + 15 (2 instructions, the per-lane offsets of a full tile against a base moved back by 16 w rows).
__device__ __forceinline__ void ntp_stage(const GemmBf16Params& p, const NtpTile& t, int stage, int kt, unsigned lds0, int wave, const unsigned (&oa)[4],
                                          const unsigned (&ob)[4]) {
    const unsigned la = lds0 + stage * B_STAGE + (wave * 32) * 128;
    // leading dimensions: 32-bit (the launcher admits ld < 2^22 only) - the kernel sits at the SGPR limit
    const char* ak = ntp_uniform((const char*)p.A + ((int64_t)t.m0 * (int)p.lda + kt * KSTEP) * 2);
    const char* bk = ntp_uniform((const char*)p.B + ((int64_t)t.n0 * (int)p.ldb + kt * KSTEP) * 2);
    if (t.half) {
        const char* akh = ak - (int64_t)(wave * 16) * (int)p.lda * 2;
        const unsigned lah = lds0 + stage * B_STAGE + (wave * 16) * 128;
        glds16_saddr(akh, oa[0], lah);
        glds16_saddr(akh, oa[1], lah + 1024);
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16_saddr(ak, oa[q], la + q * 1024);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16_saddr(bk, ob[q], la + B_A + q * 1024);
}

// One tile: main loop on the stages requested earlier, request of the next tile's first two K-tiles between the two sub-steps of the
// last K-tile, epilogue.  HALF: wave (wm, wn) owns rows 64 wm .. 64 wm + 63 (4 MFMA row tiles, accumulators a0 .. a63).
template <int EPI, bool HALF, bool STAMP>
__device__ __forceinline__ void ntp_run_tile(char* smem, unsigned lds0, int lane, int wave, int nk, const NtpTile& cur,
                                             bool first, const unsigned (&oa)[4], const unsigned (&ob)[4], int vb, unsigned* tickets, NtpTile& nxt, bool& more,
                                             int& vn) {
    constexpr int NG = HALF ? 4 : 8;
    constexpr bool OUT_F32 = EPI == NTP_RESID;
    // vector-memory operations every wave issues in the epilogue of a tile, at least: per 16-row tile two 8-row groups with one (bias, GELU')
    // or two (GELU + pre-activation, fp32) 16-byte stores and, for GELU' / the residual, as many operand loads.  A half tile may follow a
    // whole one, never the other way round, so the half tile's count serves both.  The counter holds 6 bits.
    constexpr int NV_FULL = EPI == NTP_BIAS ? 16 : EPI == NTP_RESID ? 60 : 32;
    constexpr int NV = HALF ? (EPI == NTP_BIAS ? 8 : EPI == NTP_RESID ? 32 : 16) : NV_FULL;
    const int wm = wave >> 2, wn = wave & 3;
    const GemmBf16Params& p = ntp_args();   // main loop and tile walk: A, B, lda, ldb, tiles_*, group_*, half_*
    bf16x8 a[8], b0[4], b1[4];
    const int frow = lane & 15, fq = lane >> 4;
    const unsigned pos0 = ((fq) ^ (lane & 7)) * 16, pos1 = ((4 + fq) ^ (lane & 7)) * 16;
    const unsigned fa = lds0 + (wm * (HALF ? 64 : 128) + frow) * 128, fb = lds0 + B_A + (wn * 64 + frow) * 128;
    const unsigned fa00 = fa + pos0, fa01 = fa + pos1, fa10 = fa + B_STAGE + pos0, fa11 = fa + B_STAGE + pos1;
    const unsigned fb00 = fb + pos0, fb01 = fb + pos1, fb10 = fb + B_STAGE + pos0, fb11 = fb + B_STAGE + pos1;
    // dynamic tile order: wave 0 requests the next tile's ticket with the LDS-DMA of K-tile 0 and publishes it behind K-tile 1's wait
    // (the two K-tiles of one trip of the loop below, so the register is not carried around the loop); nk >= 3, checked by the launcher
    const unsigned box = lds0 + 2 * B_STAGE;
#define NTP_FULL(S, KT, A_S1, B_S1, A_N0, B_N0, HOOK, Z)   /* K-tile KT is not the last one: as NTB_FULL; Z: the tile's first k-step */    \
    do {                                                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
        ntb_substep<0, true, false, NG, Z>(a, b0, b1, A_S1, B_S1);                                    \
        if ((S) == 0 && (KT) == 0 && !first) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(NV) : "memory");   /* K-tile 1 is older than the previous epilogue's stores */ \
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                              \
        if ((HOOK) && (S) == 1 && tickets && wave == 0 && (KT) == 1) ntp_ticket_publish(box, ticket); \
        __builtin_amdgcn_s_barrier();                                                                 \
        if ((KT) + 2 < nk) ntp_stage(p, cur, S, (KT) + 2, lds0, wave, oa, ob);                    \
        if ((HOOK) && (S) == 0 && tickets && wave == 0 && (KT) == 0) ntp_ticket_request(tickets, ticket); \
        ntb_substep<0, true, false, NG>(a, b1, b0, A_N0, B_N0);                                       \
    } while (0)
    // The accumulators start from the constant 0 of the first k-step's MFMAs (peeled first trip below); tiles of one or two K-tiles
    // clear them by hand
    const bool peel = nk > 2;
    if (!peel) ntb_zero<0>();
    // K-tile 0 of this tile: requested in the prologue (first tile: K-tile 1 is the youngest request and may stay in flight)
    // or in front of the previous tile's epilogue (its loads and stores are younger, so everything is waited for)
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
    if constexpr (STAMP) ts0 = ntp_stamp();
    if (first) {
        if (nk > 1 && !HALF) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // The stores (and operand loads) of the previous tile's epilogue were issued AFTER this tile's K-tiles 0 and 1 and the counter
        // retires in issue order: with at most NV operations outstanding both K-tiles have landed while the youngest stores are still
        // on their way to memory - they drain under K-tile 0 and the first half of K-tile 1 instead of holding the tile's start
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NV) : "memory");
    }
    __builtin_amdgcn_s_barrier();
    if constexpr (STAMP) ts1 = ntp_stamp();
    // same issue order as inside a sub-step (a0 b0 a1 b1 a2 b2 a3 b3 [a4 a5 a6 a7]): the loop's waits rely on it
    ntb_read<0>(a[0], fa00); ntb_read<0>(b0[0], fb00); ntb_read<2048>(a[1], fa00); ntb_read<2048>(b0[1], fb00);
    ntb_read<4096>(a[2], fa00); ntb_read<4096>(b0[2], fb00); ntb_read<6144>(a[3], fa00); ntb_read<6144>(b0[3], fb00);
    if constexpr (!HALF) {
        ntb_read<8192>(a[4], fa00); ntb_read<10240>(a[5], fa00); ntb_read<12288>(a[6], fa00); ntb_read<14336>(a[7], fa00);
    }
    int kt = 0;
    if (peel) {   // the first trip: K-tiles 0 and 1 (ticket request / publish), the accumulators written, not accumulated, by the first k-step
        int ticket;
        NTP_FULL(0, 0, fa01, fb01, fa10, fb10, 1, true);
        NTP_FULL(1, 1, fa11, fb11, fa00, fb00, 1, false);
        kt = 2;
    }
    for (; kt + 2 < nk; kt += 2) {
        int ticket = 0;   // kt >= 2: no ticket traffic
        NTP_FULL(0, kt, fa01, fb01, fa10, fb10, 0, false);
        NTP_FULL(1, kt + 1, fa11, fb11, fa00, fb00, 0, false);
        (void)ticket;
    }
    const bool two_left = kt + 1 < nk;
    if (two_left) {
        int ticket = 0;   // no ticket traffic in this K-tile (tickets need nk >= 3, and then kt >= 2 here)
        NTP_FULL(0, kt, fa01, fb01, fa10, fb10, 0, false);
        (void)ticket;
    }
    // last K-tile: sub-step 0 under the reads of sub-step 1; then every read of both stages has retired on every wave (barrier),
    // so the next tile's first two K-tiles are requested here, under the last MFMAs and the epilogue
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ntb_substep<0, true, false, NG>(a, b0, b1, two_left ? fa11 : fa01, two_left ? fb11 : fb01);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // the next tile: workgroup b's static list (b, b + G, ...) or, dynamic order, the ticket wave 0 published in front of this K-tile
    // (XCD x draws tile 8 k + x)
    vn = tickets ? __builtin_amdgcn_readfirstlane(*(const int*)(smem + 2 * B_STAGE)) * 8 + (int)(blockIdx.x & 7) : vb + (int)gridDim.x;
    more = vn < p.tiles_m * p.tiles_n + p.half_tiles;
    if (more) nxt = ntp_tile_of(p, vn, p.tiles_m * p.tiles_n);
    if (tickets) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // every wave has read the word before wave 0's epilogue may overwrite it (it lives in its slab)
    }
    const GemmBf16Params& pe = ntp_args();   // the epilogue's arguments (C, ldc, bias, aux pointers): requested here, under the last MFMAs
    const EpiParams& e = pe.epi;
    const float* const e_bias = e.bias;
    char* const e_c = (char*)pe.C;
    const int e_ldc = (int)pe.ldc;
    if (more) {
        ntp_stage(p, nxt, 0, 0, lds0, wave, oa, ob);
        if (nk > 1) ntp_stage(p, nxt, 1, 1, lds0, wave, oa, ob);
    }
    ntb_substep<0, false, false, NG>(a, b1, b0, 0u, 0u);
    asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");   // the last MFMAs retire before the accumulators are read back
#undef NTP_FULL
    if constexpr (STAMP) ts2 = ntp_stamp();

    // ---- epilogue of the tile
    {
        const int erow = lane >> 3, ecol = (lane & 7) * (OUT_F32 ? 4 : 8);   // fp32: columns 4 ep .. and 32 + 4 ep .. (ntp_slab_unit)
        const int mw = cur.m0 + wm * (HALF ? 64 : 128), nw = cur.n0 + wn * 64;
        NtpEpi c;
        constexpr int ES = OUT_F32 ? 4 : 2;
        c.row_bytes_c = (unsigned)(e_ldc * ES);
        c.c = e_c + ((int64_t)mw * e_ldc + nw) * ES;
        c.voff_c = (unsigned)((erow * e_ldc + ecol) * ES);
        c.aux = nullptr; c.pre = nullptr; c.row_bytes_aux = c.row_bytes_pre = 0; c.voff_aux = c.voff_pre = 0;
        if constexpr (EPI == NTP_DGELU) {
            c.row_bytes_aux = (unsigned)(e_ldc * 2);
            c.aux = (const char*)e.dgelu_pre + ((int64_t)mw * e_ldc + nw) * 2;
            c.voff_aux = (unsigned)((erow * e_ldc + ecol) * 2);
        }
        if constexpr (EPI == NTP_RESID) {
            c.row_bytes_aux = (unsigned)(e_ldc * 4);
            c.aux = (const char*)e.resid + ((int64_t)mw * e_ldc + nw) * 4;
            c.voff_aux = (unsigned)((erow * e_ldc + ecol) * 4);
        }
        if constexpr (EPI == NTP_GELU_PRE) {
            c.row_bytes_pre = (unsigned)(e_ldc * 2);
            c.pre = (char*)e.pre_out + ((int64_t)mw * e_ldc + nw) * 2;
            c.voff_pre = (unsigned)((erow * e_ldc + ecol) * 2);
        }
        c.slab = smem + 2 * B_STAGE + wave * P_SLAB;
#pragma unroll
        for (int j = 0; j < 4; ++j) c.slab_wr[j] = lds0 + 2 * B_STAGE + wave * P_SLAB + frow * 256 + (ntp_slab_unit<EPI>(4 * j + fq, frow) << 4);
        c.bias0 = f32x4{0.f, 0.f, 0.f, 0.f}; c.bias1 = c.bias0;
        if (e_bias) { c.bias0 = *(const f32x4*)(e_bias + nw + ecol); c.bias1 = *(const f32x4*)(e_bias + nw + ecol + (OUT_F32 ? 32 : 4)); }
        const bool do_cs = EPI == NTP_DGELU && e.cs_partial != nullptr;
        f32x4 cs0 = {0.f, 0.f, 0.f, 0.f}, cs1 = {0.f, 0.f, 0.f, 0.f};
        constexpr int AD = ntp_aux_depth<EPI>() - (STAMP && ntp_aux_depth<EPI>() > 1 ? 1 : 0);   // the diagnostic instances keep their time stamps in registers
        NtpAux<EPI> ring[AD];
        if constexpr (EPI == NTP_DGELU || EPI == NTP_RESID) ntp_aux_fill<EPI, 0, NG, AD>(ring, c);
        ntp_epilogue_rows<EPI, 0, NG, AD>(c, lane, ring, do_cs, cs0, cs1);
        if constexpr (EPI == NTP_DGELU) {
            if (do_cs) {   // the 8 row groups of a wave (lane bits 3..5) in a fixed order, then one partial row per 64-row slice of the output
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cs0[j] += __shfl_xor(cs0[j], 8, 64);  cs1[j] += __shfl_xor(cs1[j], 8, 64);
                    cs0[j] += __shfl_xor(cs0[j], 16, 64); cs1[j] += __shfl_xor(cs1[j], 16, 64);
                    cs0[j] += __shfl_xor(cs0[j], 32, 64); cs1[j] += __shfl_xor(cs1[j], 32, 64);
                }
                if (lane < 8) {   // partial rows: one per 128 output rows for full tiles, one per 64 rows for half tiles (the rows after them)
                    const int64_t slot = HALF ? (int64_t)(pe.half_m0 / 128) + (mw - pe.half_m0) / 64 : (int64_t)(mw / 128);
                    float* dst = e.cs_partial + slot * pe.N + nw + ecol;
                    *(f32x4*)dst = cs0;
                    *(f32x4*)(dst + 4) = cs1;
                }
            }
        }
    }
    if constexpr (STAMP) {
        ts3 = ntp_stamp();
        if (threadIdx.x == 0 && p.stamps) {
            unsigned long long* d = p.stamps + (size_t)vb * 4;
            d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = ts3;
        }
    }
}

// STAMP = true is a DIAGNOSTIC instance (tools/gemm_stamps.py): wave 0 records s_memtime at the start of a tile's main loop, at its
// end, and at the end of the epilogue into p.stamps (a buffer nothing else reads); the production instances contain no stamp.
template <int EPI, bool STAMP = false>
__global__ __launch_bounds__(512, 2) void gemm_bf16_nt_pers_kernel(GemmBf16Params) {   // read through ntp_args, see there
    __shared__ __attribute__((aligned(1024))) char smem[P_LDS];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const GemmBf16Params& p = ntp_args();
    const int nk = p.K / KSTEP;
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;

    asm volatile("" ::: SC_ACC_AGPRS);   // reserves a0..a127 in the kernel descriptor

    // LDS-DMA: per-lane byte offsets against a per-tile uniform base (tiles are whole: no row is clamped)
    unsigned oa[4], ob[4];
    {
        const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            oa[q] = (unsigned)(((wave * 32 + q * 8 + srow) * (int)p.lda + schunk * 8) * 2);
            ob[q] = (unsigned)(((wave * 32 + q * 8 + srow) * (int)p.ldb + schunk * 8) * 2);
        }
    }
    // Tile order.  Static (p.epi.tickets == null): workgroup b walks tiles b, b + G, b + 2G, ...  Dynamic: the workgroups of XCD x draw
    // tickets k = 0, 1, 2, ... from tickets[x] and compute tile 8 k + x - the tiles the XCD's workgroups walk statically, in the same
    // order - so a workgroup that starts late or is held up (CUs taken by another stream's kernels or by a collective) computes
    // fewer tiles instead of holding the launch back by a whole list.  The first ticket is drawn at the start (the one exposed round
    // trip, ~1 us); from then on the ticket of the next tile is
    // requested by wave 0 behind the barrier that opens a tile and awaited in front of the tile's last K-tile, a main loop later;
    // it is handed to the other waves through one LDS word that aliases the first bytes of wave 0's epilogue slab (the kernel owns
    // all 160 KiB), written and read while no epilogue is running.  (The kernel sits at the SGPR limit: the walk keeps no more
    // uniform state than the ticket pointer.)
    unsigned* const tickets = p.epi.tickets ? p.epi.tickets + (blockIdx.x & 7) : nullptr;
    int vb = blockIdx.x, vn = vb + (int)gridDim.x;
    if (tickets) {   // the first ticket: the one exposed round trip
        int* const box = (int*)(smem + 2 * B_STAGE);
        if (t == 0) *box = (int)atomicAdd(tickets, 1u);
        __syncthreads();
        vb = __builtin_amdgcn_readfirstlane(*box) * 8 + (int)(blockIdx.x & 7);
        __syncthreads();
    }
    if (vb < p.tiles_m * p.tiles_n + p.half_tiles) {
        NtpTile cur = ntp_tile_of(p, vb, p.tiles_m * p.tiles_n);
        ntp_stage(p, cur, 0, 0, lds0, wave, oa, ob);
        if (nk > 1) ntp_stage(p, cur, 1, 1, lds0, wave, oa, ob);
        bool first = true;
        for (;;) {
            bool more = false;
            NtpTile nxt = cur;
            if (cur.half) ntp_run_tile<EPI, true, STAMP>(smem, lds0, lane, wave, nk, cur, first, oa, ob, vb, tickets, nxt, more, vn);
            else ntp_run_tile<EPI, false, STAMP>(smem, lds0, lane, wave, nk, cur, first, oa, ob, vb, tickets, nxt, more, vn);
            if (!more) break;
            first = false;
            cur = nxt;
            vb = vn;
        }
    }
    if (tickets && t == 0) {   // the last workgroup to leave zeroes the tickets for the next launch on this stream (every other
        unsigned* const tk = tickets - (blockIdx.x & 7);   // workgroup has received its last ticket before it adds to the exit count)
        if (atomicAdd(tk + 8, 1u) == gridDim.x - 1) {
#pragma unroll
            for (int i = 0; i < 9; ++i) tk[i] = 0u;
        }
    }
}

// ------------------------------------------------------------------------------------------------ TN
// LDS tile: [64 r][128 cols] bf16, 256-B rows.  32-B slot swizzle so that the 8 rows a 32-lane half touches in one
// ds_read_b64_tr_b16 fall on 8 different 32-B slots of the 256-B bank row.
__device__ __forceinline__ int tn_swz(int row) { return ((row & 3) | (((row >> 3) & 1) << 2)) << 1; }

// Weight gradient of the implicit 3x3 convolution (conv_w != 0): B is the bordered NHWC input [R][cin] (ldb = cin) and output column
// n = tap * cin + c reads channel c of the row shifted by the tap, (ty - 1) (w + 2) + (tx - 1) rows away - an element offset per
// staging lane, fixed for the whole contraction.  The caller keeps w + 3 finite rows either side of the image.
__device__ __forceinline__ int tn_conv_col(int n, int cin, int w) {
    const int tap = n / cin, c = n - tap * cin;
    const int ty = (tap * 11) >> 5, tx = tap - 3 * ty;
    return ((ty - 1) * (w + 2) + (tx - 1)) * cin + c;
}

__global__ __launch_bounds__(256, 2) void gemm_bf16_tn_kernel(GemmBf16Params p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF_BYTES];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int id = xcd_remap(blockIdx.x, ntiles * p.splits);
    const int split = id / ntiles, tile = id % ntiles;
    const int m0 = (tile / p.tiles_n) * TILE, n0 = (tile % p.tiles_n) * TILE;
    const int kt_begin = split * p.k_per_split;
    const int nk_total = (p.K + KSTEP - 1) / KSTEP;
    const int kt_end = min(nk_total, kt_begin + p.k_per_split);

    // staging: one wave instruction = 4 rows x 256 B; lane -> (row l>>4, slot l&15)
    const int srow = lane >> 4;
    int acol[4], bcol[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = wave * 16 + q * 4 + srow;
        const int c = (lane & 15) ^ tn_swz(r);
        acol[q] = min(m0 + c * 8, p.M - 8);
        bcol[q] = min(n0 + c * 8, p.N - 8);
        if (p.conv_w) bcol[q] = tn_conv_col(bcol[q], (int)p.ldb, p.conv_w);
    }
    auto stage = [&](int buf, int kt) {
        char* abase = smem + buf * BUF_BYTES + (wave * 16) * 256;
        char* bbase = abase + OPER_BYTES;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t gr = min(kt * KSTEP + wave * 16 + q * 4 + srow, p.K - 1);
            glds16(p.A + gr * p.lda + acol[q], abase + q * 4 * 256);
            glds16(p.B + gr * p.ldb + bcol[q], bbase + q * 4 * 256);
        }
    };
    // rows past the contraction length must contribute zero (the clamped loads above fetched a valid row)
    auto zero_tail = [&](int buf, int kt) {
        const int valid = p.K - kt * KSTEP;
        if (valid >= KSTEP) return;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int piece = q * 256 + t;          // 2048 pieces of 16 B over both operand tiles
            const int row = (piece & 1023) >> 4;
            if (row >= valid) *(uint4*)(smem + buf * BUF_BYTES + piece * 16) = uint4{0u, 0u, 0u, 0u};
        }
        __syncthreads();
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Fused bias gradient: the workgroups of the first tile column also sum their staged A tiles over r.  Thread t owns the 8
    // columns of chunk t & 15 and the 4 rows 4 (t >> 4) .. +3 of every K-tile (one ds_read_b128 per row, the staging swizzle undone).
    const bool do_cs = p.cs_out != nullptr && n0 == 0;
    const int cs_cg = t & 15, cs_rg = t >> 4;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    if (kt_begin < kt_end) {
        stage(0, kt_begin);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        zero_tail(0, kt_begin);
    }
    const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3;
    const int swz = (q4 | ((g & 1) << 2)) << 1;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        const char* abase = smem + cur * BUF_BYTES;
        const char* bbase = abase + OPER_BYTES;
        if (do_cs) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = cs_rg * 4 + q;
                const uint4 v = *(const uint4*)(abase + row * 256 + ((cs_cg ^ tn_swz(row)) << 4));
                cs[0] += __uint_as_float(v.x << 16); cs[1] += __uint_as_float(v.x & 0xffff0000u);
                cs[2] += __uint_as_float(v.y << 16); cs[3] += __uint_as_float(v.y & 0xffff0000u);
                cs[4] += __uint_as_float(v.z << 16); cs[5] += __uint_as_float(v.z & 0xffff0000u);
                cs[6] += __uint_as_float(v.w << 16); cs[7] += __uint_as_float(v.w & 0xffff0000u);
            }
        }
        bf16x8 af[2][4], bfr[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = 32 * s + 8 * g + 4 * h + q4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ca = ((wm * 64 + 16 * i) >> 3) + (pp >> 1);
                    const int cb = ((wn * 64 + 16 * i) >> 3) + (pp >> 1);
                    const bf16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(abase + row * 256 + ((ca ^ swz) << 4) + ((pp & 1) << 3)));
                    const bf16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(bbase + row * 256 + ((cb ^ swz) << 4) + ((pp & 1) << 3)));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        af[s][i][4 * h + e] = va[e];
                        bfr[s][i][4 * h + e] = vb[e];
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (kt + 1 < kt_end) stage(cur ^ 1, kt + 1);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[s][j], af[s][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < kt_end) zero_tail(cur ^ 1, kt + 1);
    }

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + 16 * i + i16;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + 16 * j + 4 * g;
            if (n >= p.N) continue;
            if (p.partial) {
                *(f32x4*)(p.partial + ((int64_t)split * p.M + m) * p.N + n) = acc[i][j];
            } else {
                float* cp = (float*)p.C + (int64_t)m * p.ldc + n;
                f32x4 v = acc[i][j] * p.epi.alpha;
                if (p.epi.beta != 0.f) v += *(const f32x4*)cp * p.epi.beta;
                *(f32x4*)cp = v;
            }
        }
    }
    if (do_cs) {   // fixed-order reduction: 4 row groups per wave (lane bits 4, 5), then the 4 waves through LDS
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            cs[e] += __shfl_xor(cs[e], 16, 64);
            cs[e] += __shfl_xor(cs[e], 32, 64);
        }
        float* red = (float*)smem;   // the K loop ended on a barrier: the operand tiles are dead
        if (lane < 16) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[wave * 128 + lane * 8 + e] = cs[e];
        }
        __syncthreads();
        const int m = m0 + t;
        if (t < 128 && m < p.M) {
            const float v = (red[t] + red[128 + t]) + (red[256 + t] + red[384 + t]);
            if (p.cs_partial) p.cs_partial[(int64_t)split * p.M + m] = v;
            else p.cs_out[m] = p.cs_beta != 0.f ? v + p.cs_beta * p.cs_out[m] : v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ TN, 256x256, AGPR accumulators
// The weight-gradient GEMM with the structure of the NT 256x256 kernel above: 8 waves as 2(M) x 4(N), each wave a 128x64
// sub-tile whose 128 accumulator registers are the physical AGPRs a0..a127 (same tile -> register map, same hand-placed
// MFMA / LDS-read order).  Both operands are contraction-strided, so a K-tile is 64 rows r of 256 columns (512 B), staged
// row-major with the 32-B slot swizzle of tn_swz and read back transposed: one MFMA operand = two ds_read_b64_tr_b16.
// LDS: [A stage 0 | A stage 1 | B stage 0 | B stage 1], 32 KiB each, so that both stages of an operand are reachable from one
// per-lane base register through the 16-bit DS offset.  The contraction is split over workgroups (one round of <= 256
// workgroups); R must be a multiple of 64.  (The fused column sums of A live in the 128x128 kernel only: with the accumulators
// in AGPRs this kernel has 128 VGPRs, and 8 running sums + 16 staging registers do not fit next to 64 fragment registers;
// the launcher runs the separate column-sum pass instead.)
constexpr int TB_OPER = 32768;   // one operand tile: 64 rows x 512 B

// Fragments are kept as four dwords: the two 8-byte halves of an operand are joined at dword granularity (a pure register
// sequence).  With 16-bit element vectors the join is lowered to v_bfi/v_perm read-modify-writes ON the destination registers,
// which race with the LDS return of the very reads that fill them.
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
template <int OFF, int XOR>
__device__ __forceinline__ i32x4 tnb_read(unsigned base) {   // fragment at (base ^ XOR) + OFF: rows .. and rows + 4 (2048 B further)
    i32x2 lo, hi;
    if constexpr (XOR == 0) {
        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                     : "=&v"(lo), "=&v"(hi) : "v"(base), "n"(OFF), "n"(OFF + 2048) : "memory");
    } else {   // the XOR is done inside the statement: left to the compiler, the 12 fragment addresses are hoisted into 12 VGPRs
        unsigned tmp;
        asm volatile("v_xor_b32 %2, %5, %3\n\tds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %2 offset:%6"
                     : "=&v"(lo), "=&v"(hi), "=&v"(tmp) : "v"(base), "n"(OFF), "n"(XOR), "n"(OFF + 2048) : "memory");
    }
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
template <int F>
__device__ __forceinline__ void tnb_mfma(const i32x4 (&a)[8], const i32x4 (&b)[4]) {
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%2:%3], %0, %1, a[%2:%3]" : : "v"(b[F % 4]), "v"(a[F / 4]), "n"(4 * F), "n"(4 * F + 3) : SC_ACC_AGPRS);
}
// one sub-step (32 rows of the contraction): 8 groups of 4 MFMAs; when LOAD, group I is followed by the reads that reload
// a[I] for the next sub-step in place and, for I < 4, B fragment I of the next sub-step into the other B set.
// OFF = byte offset of the next sub-step's rows inside the operand's two-stage area.
template <int I, bool LOAD, int OFF>
__device__ __forceinline__ void tnb_substep(i32x4 (&a)[8], const i32x4 (&b)[4], i32x4 (&bn)[4], unsigned ab0, unsigned bb0) {
    tnb_mfma<4 * I>(a, b); tnb_mfma<4 * I + 1>(a, b); tnb_mfma<4 * I + 2>(a, b); tnb_mfma<4 * I + 3>(a, b);
    if constexpr (LOAD) {
        a[I] = tnb_read<OFF, (I << 5)>(ab0);          // fragment I: 16-byte slot index ^ 2 I (see the base computation)
        if constexpr (I < 4) bn[I] = tnb_read<OFF, (I << 5)>(bb0);
    }
    if constexpr (I + 1 < 8) tnb_substep<I + 1, LOAD, OFF>(a, b, bn, ab0, bb0);
}
template <int T>
__device__ __forceinline__ void tnb_store(const GemmBf16Params& p, int split, int mrow, int ncol) {   // tile T = 4 i + j of the wave
    const int m = mrow + 16 * (T / 4), n = ncol + 16 * (T % 4);
    if (m < p.M && n < p.N) {
        const f32x4 acc = ntb_acc<T>();
        if (p.partial) {
            *(f32x4*)(p.partial + ((int64_t)split * p.M + m) * p.N + n) = acc;
        } else {
            float* cp = (float*)p.C + (int64_t)m * p.ldc + n;
            f32x4 v = acc * p.epi.alpha;
            if (p.epi.beta != 0.f) v += *(const f32x4*)cp * p.epi.beta;
            *(f32x4*)cp = v;
        }
    }
    if constexpr (T + 1 < 32) tnb_store<T + 1>(p, split, mrow, ncol);
}

// the work of ONE workgroup: output tile `id % ntiles`, contraction slice `id / ntiles` of problem p
__device__ __forceinline__ void tn_big_body(const GemmBf16Params& p, char* smem, int id) {
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int split = id / ntiles, tile = id % ntiles;
    const int m0 = (tile / p.tiles_n) * B_M, n0 = (tile % p.tiles_n) * B_N;
    const int kt_begin = split * p.k_per_split;
    const int kt_end = min(p.K / KSTEP, kt_begin + p.k_per_split);
    const int nk = kt_end - kt_begin;

    // staging: one wave instruction = 2 rows x 512 B; lane -> (row l >> 5, slot l & 31); wave w owns rows 8 w .. 8 w + 7.
    // Per-lane 32-bit byte offsets against a uniform (SGPR) base that advances by one K-tile: 8 VGPRs instead of 8 pointers.
    unsigned oa[4], ob[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = wave * 8 + q * 2 + (lane >> 5);
        const int c = (lane & 31) ^ tn_swz(r);
        oa[q] = (unsigned)((r * p.lda + min(m0 + c * 8, p.M - 8)) * 2);
        const int nb = min(n0 + c * 8, p.N - 8);
        // implicit convolution: the tap's row shift rides in the lane offset, kept non-negative by starting the base w + 3 rows early
        ob[q] = p.conv_w ? (unsigned)(((r + p.conv_w + 3) * (int)p.ldb + tn_conv_col(nb, (int)p.ldb, p.conv_w)) * 2) : (unsigned)((r * p.ldb + nb) * 2);
    }
    const char* a_tile0 = (const char*)p.A + (int64_t)kt_begin * KSTEP * p.lda * 2;
    const char* b_tile0 = (const char*)p.B + ((int64_t)kt_begin * KSTEP - (p.conv_w ? p.conv_w + 3 : 0)) * p.ldb * 2;
    const int64_t a_step = (int64_t)KSTEP * p.lda * 2, b_step = (int64_t)KSTEP * p.ldb * 2;   // bytes per K-tile

    asm volatile("" ::: SC_ACC_AGPRS);   // reserve the accumulator AGPRs in the kernel descriptor

    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3;
    const int swz = (q4 | ((g & 1) << 2)) << 1;           // == tn_swz(row) for every row this lane reads
    // Fragment i of A: slot (wm 16 + 2 i + (pp >> 1)) ^ swz of row 8 g + q4 (+ 32 s + 4 h).  swz only has bits 1..3, 2 i too, and
    // the LDS array starts on a 1-KiB boundary, so fragment i sits at (base of fragment 0) ^ (i << 5): one base register per operand.
    const unsigned ab0 = lds0 + (8 * g + q4) * 512 + (((wm * 16 + (pp >> 1)) ^ swz) << 4) + ((pp & 1) << 3);
    const unsigned bb0 = lds0 + 2 * TB_OPER + (8 * g + q4) * 512 + (((wn * 8 + (pp >> 1)) ^ swz) << 4) + ((pp & 1) << 3);

#define TNB_STAGE(S, KT)                                                                              \
    do {                                                                                              \
        const unsigned la__ = lds0 + (S) * TB_OPER + (wave * 8) * 512;                                \
        const char* ak__ = a_tile0 + (KT) * a_step;                                                   \
        const char* bk__ = b_tile0 + (KT) * b_step;                                                   \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) glds16_saddr(ak__, oa[q], la__ + q * 1024);     \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) glds16_saddr(bk__, ob[q], la__ + 2 * TB_OPER + q * 1024); \
    } while (0)
    // The fragment registers are written by inline-assembly LDS reads the compiler knows nothing about: between such a read and
    // its s_waitcnt no compiler-scheduled vector code may touch them (audited at build time, _asm_check.py).
    // Stage S holds K-tile KT (relative to kt_begin).  Same pipeline as the NT kernel: sub-step 0 || reads of sub-step 1 (stage S),
    // barrier (tile KT+1 landed, every read of stage S retired), LDS-DMA of tile KT+2 into stage S, sub-step 1 || reads of the
    // first fragments of tile KT+1 (stage S^1).
#define TNB_FULL(S, KT)   /* tile KT is not the last one of this workgroup */                          \
    do {                                                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
        tnb_substep<0, true, (S) * TB_OPER + 16384>(a, b0, b1, ab0, bb0);                             \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                   \
        __builtin_amdgcn_s_barrier();                                                                 \
        if ((KT) + 2 < nk) TNB_STAGE(S, (KT) + 2);                                                    \
        tnb_substep<0, true, ((S) ^ 1) * TB_OPER>(a, b1, b0, ab0, bb0);                               \
    } while (0)
#define TNB_LAST(AB, BB)   /* the stage is a run-time offset folded into the bases */                  \
    do {                                                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
        tnb_substep<0, true, 16384>(a, b0, b1, AB, BB);                                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
        tnb_substep<0, false, 0>(a, b1, b0, AB, BB);                                                  \
    } while (0)

    i32x4 a[8], b0[4], b1[4];
    TNB_STAGE(0, 0);   // as in the NT kernel: both stages requested up front, accumulators cleared under the latency
    if (nk > 1) TNB_STAGE(1, 1);
    ntb_zero<0>();
    if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    b0[0] = tnb_read<0, 0>(bb0); b0[1] = tnb_read<0, 32>(bb0); b0[2] = tnb_read<0, 64>(bb0); b0[3] = tnb_read<0, 96>(bb0);
    a[0] = tnb_read<0, 0>(ab0); a[1] = tnb_read<0, 32>(ab0); a[2] = tnb_read<0, 64>(ab0); a[3] = tnb_read<0, 96>(ab0);
    a[4] = tnb_read<0, 128>(ab0); a[5] = tnb_read<0, 160>(ab0); a[6] = tnb_read<0, 192>(ab0); a[7] = tnb_read<0, 224>(ab0);
    int kt = 0;   // as in the NT kernel: every loop exit leads to code that starts with a wait
    for (; kt + 2 < nk; kt += 2) {
        TNB_FULL(0, kt);
        TNB_FULL(1, kt + 1);
    }
    const bool two_left = kt + 1 < nk;
    if (two_left) TNB_FULL(0, kt);
    const unsigned last_stage = two_left ? TB_OPER : 0;
    TNB_LAST(ab0 + last_stage, bb0 + last_stage);
    asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");   // no LDS read in flight on any exit edge; the last MFMAs retire before the accumulators are read back
#undef TNB_LAST
#undef TNB_FULL
#undef TNB_STAGE

    // D[n][m]: lane holds m = .. + (lane & 15), n = .. + 4 (lane >> 4) + reg
    tnb_store<0>(p, split, m0 + wm * 128 + i16, n0 + wn * 64 + 4 * g);
}

__global__ __launch_bounds__(512, 2) void gemm_bf16_tn_big_kernel(GemmBf16Params p) {
    __shared__ __attribute__((aligned(1024))) char smem[4 * TB_OPER];
    tn_big_body(p, smem, xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n * p.splits));
}

// Up to four weight-gradient problems with one contraction length (the four dW of a transformer block: same rows, different dY / X)
// in ONE launch: the output tiles of all problems share the chip, so the contraction is split 2-5 ways instead of 7-28 ways per
// problem - a few large partial slabs and one reduce launch per block instead of four of each.
struct TnProb {
    const bf16_t* A; const bf16_t* B; float* C; float* partial;
    int M, N, tiles_m, tiles_n;
    int64_t lda, ldb, ldc;
    int first;   // first workgroup of the problem
};
struct TnGroup {
    TnProb prob[4];
    int n, total, K, splits, k_per_split;
    float alpha, beta;
};
__global__ __launch_bounds__(512, 2) void gemm_bf16_tn_group_kernel(TnGroup g) {
    __shared__ __attribute__((aligned(1024))) char smem[4 * TB_OPER];
    const int id = xcd_remap(blockIdx.x, g.total);
    int k = 0;
    if (g.n > 1 && id >= g.prob[1].first) k = 1;
    if (g.n > 2 && id >= g.prob[2].first) k = 2;
    if (g.n > 3 && id >= g.prob[3].first) k = 3;
    const TnProb& q = g.prob[k];
    GemmBf16Params p;
    p.conv_w = p.conv_h = p.conv_kpt_log2 = 0;
    p.A = q.A; p.B = q.B; p.C = q.C; p.partial = q.partial;
    p.M = q.M; p.N = q.N; p.K = g.K; p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc;
    p.tiles_m = q.tiles_m; p.tiles_n = q.tiles_n; p.splits = g.splits; p.k_per_split = g.k_per_split;
    p.epi.alpha = g.alpha; p.epi.beta = g.beta;
    tn_big_body(p, smem, id - q.first);
}
// C_k = alpha * sum_s partial_k[s] + beta * C_k for every problem of the group (fixed order), one launch
__global__ __launch_bounds__(256) void splitk_reduce_group_kernel(TnGroup g, int4 first_block) {
    int k = 0;
    const int b = blockIdx.x;
    if (g.n > 1 && b >= first_block.y) k = 1;
    if (g.n > 2 && b >= first_block.z) k = 2;
    if (g.n > 3 && b >= first_block.w) k = 3;
    const int fb = k == 0 ? first_block.x : (k == 1 ? first_block.y : (k == 2 ? first_block.z : first_block.w));
    const TnProb& q = g.prob[k];
    const int64_t mn = (int64_t)q.M * q.N;
    const int64_t i4 = ((int64_t)(b - fb) * 256 + threadIdx.x) * 4;
    if (i4 >= mn) return;
    f32x4 s = *(const f32x4*)(q.partial + i4);
    for (int sidx = 1; sidx < g.splits; ++sidx) s += *(const f32x4*)(q.partial + (int64_t)sidx * mn + i4);
    float* cp = q.C + (i4 / q.N) * q.ldc + (i4 % q.N);
    f32x4 v = s * g.alpha;
    if (g.beta != 0.f) v += *(const f32x4*)cp * g.beta;
    *(f32x4*)cp = v;
}

// C = alpha * sum_s partial[s] + beta * C   (fixed order); optionally also cs_out = cs_beta * cs_out + sum_s cs_partial[s].
// SL = slices of the split range per output quad (threads t % SL of a block; combined through LDS in slice order): a one-tile
// weight gradient split 256 ways has 16 K outputs and 16 MB of partials - with one thread per quad 16 workgroups would read them.
template <int SL>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* partial, int splits, int64_t mn, int n, float* c, int64_t ldc,
                                                            float alpha, float beta, const float* cs_partial, float* cs_out, int m_len, float cs_beta) {
    __shared__ f32x4 sm[SL > 1 ? 256 : 1];
    constexpr int QPB = 256 / SL;      // output quads per block
    const int sl = threadIdx.x / QPB, q = threadIdx.x % QPB;
    const int64_t i4 = ((int64_t)blockIdx.x * QPB + q) * 4;
    const bool live = i4 < mn;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (live)
        for (int k = sl; k < splits; k += SL) s += *(const f32x4*)(partial + (int64_t)k * mn + i4);
    if (SL > 1) {
        sm[threadIdx.x] = s;
        __syncthreads();
        if (sl != 0) return;
        for (int j = 1; j < SL; ++j) s += sm[j * QPB + q];
    }
    if (!live) return;
    float* cp = c + (i4 / n) * ldc + (i4 % n);
    f32x4 v = s * alpha;
    if (beta != 0.f) v += *(const f32x4*)cp * beta;
    *(f32x4*)cp = v;
    if (cs_partial && i4 < m_len) {
        f32x4 b = *(const f32x4*)(cs_partial + i4);
        for (int k = 1; k < splits; ++k) b += *(const f32x4*)(cs_partial + (int64_t)k * m_len + i4);
        if (cs_beta != 0.f) b += *(const f32x4*)(cs_out + i4) * cs_beta;
        *(f32x4*)(cs_out + i4) = b;
    }
}

// TN kernel choice.  Default: the 256x256 AGPR kernel when the output has at least 6 such tiles and R % 64 == 0, otherwise the
// 128x128 2-stage kernel at 2 workgroups/CU.  SC_GEMM_TN=128 forces the latter everywhere (run once by the test-suite).
enum TnKind { TN_SMALL = 0, TN_BIG = 2 };
int tn_env() {
    static const int v = [] { const char* e = getenv("SC_GEMM_TN"); return (e && e[0] == '1') ? 128 : 0; }();
    return v;
}
// Contraction splits of the TN kernels: enough workgroups to fill the chip even for a one-tile weight gradient over millions of rows (a
// ResNet stem / first-stage convolution: [32..64] x [64..576] outputs, 0.8 - 3.2 M rows - HBM-bound, so every CU has to stream).
constexpr int TN_MAX_SPLITS = 256;
void tn_plan_small(int64_t m, int64_t n, int64_t r, int& kind, int& splits) {
    const int64_t nk = sc_cdiv(r, KSTEP);
    int64_t s = sc_cdiv(768, sc_cdiv(m, TILE) * sc_cdiv(n, TILE));
    const int64_t cap = nk / 4 > 1 ? nk / 4 : 1;
    if (s > cap) s = cap;
    if (s > TN_MAX_SPLITS) s = TN_MAX_SPLITS;
    kind = TN_SMALL;
    splits = (int)(s < 1 ? 1 : s);
}
void tn_plan(int64_t m, int64_t n, int64_t r, bool colsum, int& kind, int& splits) {
    const int64_t nk = sc_cdiv(r, KSTEP);
    const int64_t tiles_big = sc_cdiv(m, B_M) * sc_cdiv(n, B_N);
    int64_t s;
    if (tn_env() == 0 && r % KSTEP == 0 && tiles_big >= 6) {
        kind = TN_BIG;
        s = 256 / tiles_big;       // one round of at most 256 workgroups
    } else {
        kind = TN_SMALL;
        s = sc_cdiv(768, sc_cdiv(m, TILE) * sc_cdiv(n, TILE));
    }
    const int64_t cap = nk / 4 > 1 ? nk / 4 : 1;
    if (s > cap) s = cap;
    if (s > TN_MAX_SPLITS) s = TN_MAX_SPLITS;
    splits = (int)(s < 1 ? 1 : s);
}

unsigned long long* g_nt_stamps = nullptr;   // diagnostic only (sc_gemm_bf16_nt_stamps); not part of the production path

// CUs of the current device, rounded down to a multiple of 8 (the persistent kernel's grid must keep "blocks b and b + 8 share an
// XCD" aligned with its virtual block ids)
int sc_num_cus() {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
        const char* e = sc_debug_env("SC_GEMM_CUS");   // A/B: persistent grids of fewer workgroups (round 3: 256 -> 128 CUs per launch changes the step by +2 %)
        if (e && atoi(e) >= 8) cus = atoi(e);
        return (cus / 8) * 8;
    }();
    return n;
}

}  // namespace

int sc_gemm_bf16_nt_launch(int64_t m, int64_t n, int64_t k, const void* a, int64_t lda, const void* b, int64_t ldb, void* c,
                           int64_t ldc, int out_dtype, const EpiParams& epi, hipStream_t stream) {
    SC_REQUIRE(m > 0 && n > 0 && k > 0, SC_ERR_SHAPE, "sc_gemm_bf16_nt: empty problem");
    SC_REQUIRE(a && b && c, SC_ERR_ARG, "sc_gemm_bf16_nt: null operand");
    SC_REQUIRE(k % KSTEP == 0, SC_ERR_SHAPE, "sc_gemm_bf16_nt: K = %lld must be a multiple of 64", (long long)k);
    SC_REQUIRE(n % 4 == 0, SC_ERR_SHAPE, "sc_gemm_bf16_nt: N = %lld must be a multiple of 4", (long long)n);
    SC_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && lda >= k && ldb >= k && ldc >= n, SC_ERR_SHAPE, "sc_gemm_bf16_nt: bad leading dimension");
    SC_REQUIRE(sc_aligned(a, 16) && sc_aligned(b, 16) && sc_aligned(c, 16), SC_ERR_ALIGN, "sc_gemm_bf16_nt: operands must be 16-byte aligned");
    SC_REQUIRE(out_dtype == SC_BF16 || out_dtype == SC_F32, SC_ERR_DTYPE, "sc_gemm_bf16_nt: bad out dtype");
    SC_REQUIRE(!(epi.pre_out || epi.resid || epi.dgelu_pre) || epi.ld_aux % 4 == 0, SC_ERR_SHAPE, "sc_gemm_bf16_nt: ld_aux must be a multiple of 4");
    SC_REQUIRE(epi.beta == 0.f || out_dtype == SC_F32, SC_ERR_ARG, "sc_gemm_bf16_nt: beta needs an fp32 C");
    SC_REQUIRE(!epi.colsum || (epi.colsum_ws && n % 8 == 0), SC_ERR_ARG, "sc_gemm_bf16_nt: colsum needs a workspace and N % 8 == 0");
    GemmBf16Params p;
    p.conv_w = p.conv_h = p.conv_kpt_log2 = 0;
    p.A = (const bf16_t*)a; p.B = (const bf16_t*)b; p.C = c;
    p.M = (int)m; p.N = (int)n; p.K = (int)k;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.splits = 1; p.k_per_split = 0; p.partial = nullptr;
    p.stamps = g_nt_stamps;
    p.epi = epi;
    p.epi.cs_partial = nullptr;
    bool cs_fused = false;
    int64_t cs_rows_done = 0;   // rows whose column sums the fused path covers
    static const bool small_tile = [] { const char* e = getenv("SC_GEMM_NT"); return e && e[0] == '1'; }();   // SC_GEMM_NT=128: A/B runs
    if (small_tile || n % 8 != 0 || ldc % 8 != 0 || (epi.ld_aux % 8 != 0 && (epi.pre_out || epi.resid || epi.dgelu_pre))) {
        p.tiles_m = (int)sc_cdiv(m, TILE); p.tiles_n = (int)sc_cdiv(n, TILE);
        const unsigned grid = (unsigned)(p.tiles_m * p.tiles_n);
        if (out_dtype == SC_F32) hipLaunchKernelGGL(gemm_bf16_nt_kernel<true>, dim3(grid), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(gemm_bf16_nt_kernel<false>, dim3(grid), dim3(256), 0, stream, p);
    } else {
        p.tiles_m = (int)sc_cdiv(m, T_M); p.tiles_n = (int)sc_cdiv(n, T_N);
        const unsigned grid = (unsigned)(p.tiles_m * p.tiles_n);
        static const int variant = [] { const char* e = getenv("SC_GEMM_NT"); return (e && e[0] == '2') ? 2 : 0; }();   // SC_GEMM_NT=2: 256x128 kernel everywhere (run once by the test-suite)
        static const int big_min_n = [] { const char* e = sc_debug_env("SC_GEMM_NT_BIG_MINN"); return e ? atoi(e) : 512; }();   // A/B knob (1536: wide outputs only)
        if (variant == 0 && n >= big_min_n && m >= 4096) {
            // Tile-count quantisation: all tiles cost the same, so ceil(tiles / 256) rounds are paid even when the last one is
            // nearly empty (600 tiles = 2.34 -> 3 rounds at N = 768).  When it pays, only the row tiles that fill WHOLE rounds go
            // to the 256x256 kernel and the remaining rows are a second launch of the 256x128 kernel (half-size tiles: the
            // leftover becomes one well-filled round that costs ~0.55 of a 256x256 round).  SC_GEMM_NT_SPLIT=0 disables.
            static const bool split_on = [] { const char* e = sc_debug_env("SC_GEMM_NT_SPLIT"); return !(e && e[0] == '0'); }();
            const int64_t tn_b = sc_cdiv(n, B_N), tm_b = sc_cdiv(m, B_M);
            int64_t m_main = m;
            {
                const int64_t tiles = tm_b * tn_b, full = tiles / 256, rem = tiles - full * 256;
                const bool cs_ok = !epi.colsum || epi.colsum_ws_bytes >= (size_t)1024 * n * sizeof(float);
                if (split_on && cs_ok && full >= 1 && rem > 0) {
                    const int64_t tm_main = (full * 256) / tn_b;
                    const int64_t rows_rem = m - tm_main * B_M;
                    const int64_t rounds_rem = sc_cdiv(sc_cdiv(rows_rem, T_M) * sc_cdiv(n, T_N), 256);
                    if (tm_main > 0 && rows_rem > 0 && (double)full + 0.55 * (double)rounds_rem < (double)(full + 1) - 0.1) m_main = tm_main * B_M;
                }
            }
            static const int g_m = [] { const char* e = sc_debug_env("SC_GEMM_NT_GROUP_M"); return e && atoi(e) > 0 ? atoi(e) : B_GROUP_M; }();   // A/B knobs
            static const int g_n = [] { const char* e = sc_debug_env("SC_GEMM_NT_GROUP_N"); return e && atoi(e) > 0 ? atoi(e) : B_GROUP_N; }();
            p.group_m = g_m; p.group_n = g_n;
            p.half_tiles = 0; p.half_m0 = 0;
            // persistent kernel: whole tiles and one of the step's four epilogues; SC_GEMM_NT=b forces one tile per workgroup (A/B)
            static const bool one_tile_wg = [] { const char* e = getenv("SC_GEMM_NT"); return e && e[0] == 'b'; }();
            const EpiParams& pe = p.epi;
            int kind = -1;
            const bool aux_ok = epi.ld_aux == ldc && epi.ld_aux < (1 << 22);
            const bool cs_room = epi.colsum && epi.colsum_ws_bytes >= (size_t)(m / 64 + 2) * n * sizeof(float) && sc_aligned(epi.colsum_ws, 16);
            if (!one_tile_wg && pe.alpha == 1.f && pe.beta == 0.f && m % (B_M / 2) == 0 && n % B_N == 0 && lda < (1 << 22) && ldb < (1 << 22) && ldc < (1 << 22)) {
                if (out_dtype == SC_F32) {
                    if (pe.resid && pe.resid_dtype == SC_F32 && !pe.pre_out && pe.act == 0 && !pe.dgelu_pre && !epi.colsum && aux_ok) kind = NTP_RESID;
                } else if (pe.pre_out && pe.act == 1 && !pe.resid && !pe.dgelu_pre && !epi.colsum && aux_ok) kind = NTP_GELU_PRE;
                else if (pe.dgelu_pre && !pe.pre_out && pe.act == 0 && !pe.resid && aux_ok && (!epi.colsum || cs_room)) kind = NTP_DGELU;
                else if (!pe.dgelu_pre && !pe.pre_out && pe.act == 0 && !pe.resid && !epi.colsum) kind = NTP_BIAS;
            }
            if (kind >= 0) {
                // One workgroup per CU walks the tile list: `full` 256x256 tiles, then half tiles (128x256) over the remaining rows.
                // Tile-count quantisation: with 600 tiles on 256 CUs (N = 768) a third round of whole tiles would run 88 workgroups
                // while 168 CUs idle; turning the rows of that partial round into 176 half tiles costs every CU at most half a
                // tile time (+10 %: same prologue / epilogue overheads on half the MFMA work).  The split is chosen by evaluating
                // the most loaded workgroup's cost for "all whole tiles" and for "whole rounds + half tiles".
                const int G = sc_num_cus();
                const int64_t tm_all = m / B_M, tail = (m % B_M) / (B_M / 2);        // tail = 1: a last 128-row panel
                auto cost = [&](int64_t full_tiles, int64_t half_tiles) {
                    double worst = 0.0;
                    const int64_t total = full_tiles + half_tiles;
                    for (int b = 0; b < G && b < total; ++b) {
                        const int64_t mine = (total - b + G - 1) / G;
                        const int64_t mine_full = full_tiles > b ? (full_tiles - b + G - 1) / G : 0;
                        const double c = (double)mine_full + 0.55 * (double)(mine - mine_full);
                        worst = c > worst ? c : worst;
                    }
                    return worst;
                };
                int64_t tm_main = tm_all;
                {
                    const int64_t rounds = (tm_all * tn_b) / G;
                    const int64_t tm_b2 = (rounds * G) / tn_b;      // row tiles that fill whole rounds
                    static const bool split_on2 = [] { const char* e = sc_debug_env("SC_GEMM_NT_SPLIT"); return !(e && e[0] == '0'); }();
                    if (split_on2 && tm_b2 < tm_all &&
                        cost(tm_b2 * tn_b, ((tm_all - tm_b2) * 2 + tail) * tn_b) + 0.02 < cost(tm_all * tn_b, tail * tn_b))
                        tm_main = tm_b2;
                }
                static const bool tickets_on = [] { const char* e = sc_debug_env("SC_GEMM_TICKETS"); return !(e && e[0] == '0'); }();   // =0: fixed tile lists (A/B runs)
                if (!tickets_on || k < 3 * KSTEP) p.epi.tickets = nullptr;   // the ticket protocol needs three K-tiles
                p.M = (int)m;
                p.tiles_m = (int)tm_main; p.tiles_n = (int)tn_b;
                p.half_m0 = (int)(tm_main * B_M);
                p.half_tiles = (int)(((m - tm_main * B_M) / (B_M / 2)) * tn_b);
                const int64_t total = (int64_t)p.tiles_m * p.tiles_n + p.half_tiles;
                int64_t cs_slabs = 0;
                if (epi.colsum) {   // partial rows: one per 128 output rows of the whole tiles, one per 64 rows of the half tiles; every slot is written
                    p.epi.cs_partial = (float*)epi.colsum_ws;
                    cs_slabs = tm_main * 2 + (m - tm_main * B_M) / 64;
                    cs_fused = true;
                }
                const unsigned gridp = (unsigned)(total < G ? total : G);
                if (p.stamps) {   // diagnostic instances
                    if (kind == NTP_RESID) hipLaunchKernelGGL((gemm_bf16_nt_pers_kernel<NTP_RESID, true>), dim3(gridp), dim3(512), 0, stream, p);
                    else if (kind == NTP_GELU_PRE) hipLaunchKernelGGL((gemm_bf16_nt_pers_kernel<NTP_GELU_PRE, true>), dim3(gridp), dim3(512), 0, stream, p);
                    else if (kind == NTP_DGELU) hipLaunchKernelGGL((gemm_bf16_nt_pers_kernel<NTP_DGELU, true>), dim3(gridp), dim3(512), 0, stream, p);
                    else hipLaunchKernelGGL((gemm_bf16_nt_pers_kernel<NTP_BIAS, true>), dim3(gridp), dim3(512), 0, stream, p);
                } else if (kind == NTP_RESID) hipLaunchKernelGGL(gemm_bf16_nt_pers_kernel<NTP_RESID>, dim3(gridp), dim3(512), 0, stream, p);
                else if (kind == NTP_GELU_PRE) hipLaunchKernelGGL(gemm_bf16_nt_pers_kernel<NTP_GELU_PRE>, dim3(gridp), dim3(512), 0, stream, p);
                else if (kind == NTP_DGELU) hipLaunchKernelGGL(gemm_bf16_nt_pers_kernel<NTP_DGELU>, dim3(gridp), dim3(512), 0, stream, p);
                else hipLaunchKernelGGL(gemm_bf16_nt_pers_kernel<NTP_BIAS>, dim3(gridp), dim3(512), 0, stream, p);
                SC_CHECK_LAUNCH();
                if (epi.colsum) return sc_colsum_reduce((const float*)epi.colsum_ws, (int)cs_slabs, n, epi.colsum, epi.colsum_accumulate, stream);
                return SC_OK;
            }
            // general shapes / epilogues: one 256x256 tile per workgroup, the rows of a partly filled round on the 256x128 kernel
            p.M = (int)m_main;
            p.tiles_m = (int)sc_cdiv(m_main, B_M); p.tiles_n = (int)tn_b;
            const unsigned gridb = (unsigned)(p.tiles_m * p.tiles_n);
            if (epi.colsum && epi.colsum_ws_bytes >= (size_t)2 * p.tiles_m * n * sizeof(float) && sc_aligned(epi.colsum_ws, 16)) {
                p.epi.cs_partial = (float*)epi.colsum_ws;   // [2 * tiles_m][N]; rows past M contribute nothing, every slot is written
                cs_fused = true;
            }
            if (out_dtype == SC_F32) hipLaunchKernelGGL(gemm_bf16_nt_big_kernel<true>, dim3(gridb), dim3(512), 0, stream, p);
            else hipLaunchKernelGGL(gemm_bf16_nt_big_kernel<false>, dim3(gridb), dim3(512), 0, stream, p);
            cs_rows_done = m_main;
            if (m_main < m) {   // the remaining rows: same operands and epilogue, pointers advanced by m_main rows
                GemmBf16Params q = p;
                q.epi.cs_partial = nullptr;
                q.A = p.A + m_main * lda;
                q.C = (char*)c + (size_t)m_main * ldc * (out_dtype == SC_F32 ? 4 : 2);
                q.M = (int)(m - m_main);
                if (q.epi.pre_out) q.epi.pre_out = (bf16_t*)q.epi.pre_out + m_main * epi.ld_aux;
                if (q.epi.dgelu_pre) q.epi.dgelu_pre = (const bf16_t*)q.epi.dgelu_pre + m_main * epi.ld_aux;
                if (q.epi.resid) q.epi.resid = (const char*)q.epi.resid + (size_t)m_main * epi.ld_aux * (epi.resid_dtype == SC_F32 ? 4 : 2);
                q.tiles_m = (int)sc_cdiv(m - m_main, T_M); q.tiles_n = (int)sc_cdiv(n, T_N);
                const unsigned grid2 = (unsigned)(q.tiles_m * q.tiles_n);
                if (out_dtype == SC_F32) hipLaunchKernelGGL(gemm_bf16_nt256_kernel<true>, dim3(grid2), dim3(512), 0, stream, q);
                else hipLaunchKernelGGL(gemm_bf16_nt256_kernel<false>, dim3(grid2), dim3(512), 0, stream, q);
            }
        } else {
            if (out_dtype == SC_F32) hipLaunchKernelGGL(gemm_bf16_nt256_kernel<true>, dim3(grid), dim3(512), 0, stream, p);
            else hipLaunchKernelGGL(gemm_bf16_nt256_kernel<false>, dim3(grid), dim3(512), 0, stream, p);
        }
    }
    SC_CHECK_LAUNCH();
    if (epi.colsum) {
        if (!cs_fused) return sc_colsum(c, out_dtype, m, n, ldc, epi.colsum, epi.colsum_accumulate, epi.colsum_ws, epi.colsum_ws_bytes, (void*)stream);
        SC_TRY(sc_colsum_reduce((const float*)epi.colsum_ws, 2 * p.tiles_m, n, epi.colsum, epi.colsum_accumulate, stream));
        if (cs_rows_done < m)   // rows of the second launch: a pass over that part of C, added on top (stream-ordered, so ws is free again)
            return sc_colsum((const char*)c + (size_t)cs_rows_done * ldc * (out_dtype == SC_F32 ? 4 : 2), out_dtype, m - cs_rows_done, n, ldc, epi.colsum, 1,
                             epi.colsum_ws, epi.colsum_ws_bytes, (void*)stream);
    }
    return SC_OK;
}

size_t sc_gemm_bf16_tn_ws(int64_t m, int64_t n, int64_t r) {
    int kind, s0, s1;
    int s2;
    tn_plan(m, n, r, false, kind, s0);
    tn_plan(m, n, r, true, kind, s1);
    tn_plan_small(m, n, r, kind, s2);
    const int s = s0 > s1 ? (s0 > s2 ? s0 : s2) : (s1 > s2 ? s1 : s2);
    return s > 1 ? (size_t)s * (m * n + m) * sizeof(float) : 0;   // partial C slabs + partial column sums of A
}

int sc_gemm_bf16_tn_launch(int64_t m, int64_t n, int64_t r, const void* a, int64_t lda, const void* b, int64_t ldb, float* c,
                           int64_t ldc, float alpha, float beta, void* ws, size_t ws_bytes, hipStream_t stream, float* colsum_a, float colsum_beta,
                           int conv_w) {
    SC_REQUIRE(m > 0 && n > 0 && r > 0, SC_ERR_SHAPE, "sc_gemm_bf16_tn: empty problem");
    SC_REQUIRE(a && b && c, SC_ERR_ARG, "sc_gemm_bf16_tn: null operand");
    SC_REQUIRE(m % 8 == 0 && n % 8 == 0, SC_ERR_SHAPE, "sc_gemm_bf16_tn: M and N must be multiples of 8");
    SC_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && lda >= m && (conv_w ? ldb * 9 == n : ldb >= n) && ldc >= n, SC_ERR_SHAPE,
               "sc_gemm_bf16_tn: bad leading dimension");
    SC_REQUIRE(sc_aligned(a, 16) && sc_aligned(b, 16) && sc_aligned(c, 16), SC_ERR_ALIGN, "sc_gemm_bf16_tn: operands must be 16-byte aligned");
    GemmBf16Params p;
    p.conv_w = conv_w; p.conv_h = p.conv_kpt_log2 = 0;
    p.A = (const bf16_t*)a; p.B = (const bf16_t*)b; p.C = c;
    p.M = (int)m; p.N = (int)n; p.K = (int)r;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.stamps = nullptr;
    SC_REQUIRE(!colsum_a || sc_aligned(colsum_a, 16), SC_ERR_ALIGN, "sc_gemm_bf16_tn: colsum must be 16-byte aligned");
    int kind;
    tn_plan(m, n, r, colsum_a != nullptr, kind, p.splits);
    if (kind == TN_BIG && colsum_a && (colsum_beta != 0.f && colsum_beta != 1.f)) kind = TN_SMALL;   // the separate pass only knows overwrite / accumulate
    if (kind == TN_BIG && colsum_a && ws_bytes < (size_t)1024 * m * sizeof(float)) kind = TN_SMALL;    // its partials reuse ws
    if (kind == TN_SMALL) { int k2; tn_plan_small(m, n, r, k2, p.splits); }
    const int tm = kind == TN_SMALL ? TILE : B_M, tn = kind == TN_SMALL ? TILE : B_N;
    p.tiles_m = (int)sc_cdiv(m, tm); p.tiles_n = (int)sc_cdiv(n, tn);
    p.cs_out = colsum_a; p.cs_partial = nullptr; p.cs_beta = colsum_beta;
    const int64_t nk = sc_cdiv(r, KSTEP);
    p.k_per_split = (int)sc_cdiv(nk, p.splits);
    p.splits = (int)sc_cdiv(nk, p.k_per_split);
    p.partial = nullptr;
    p.epi = epi_plain(alpha, beta);
    if (p.splits > 1) {
        SC_REQUIRE(ws && ws_bytes >= (size_t)p.splits * (m * n + m) * sizeof(float), SC_ERR_WORKSPACE, "sc_gemm_bf16_tn: workspace too small");
        SC_REQUIRE(sc_aligned(ws, 16), SC_ERR_ALIGN, "sc_gemm_bf16_tn: workspace must be 16-byte aligned");
        p.partial = (float*)ws;
        if (colsum_a) p.cs_partial = (float*)ws + (size_t)p.splits * m * n;
    }
    const unsigned grid = (unsigned)(p.tiles_m * p.tiles_n * p.splits);
    if (kind == TN_BIG) { p.cs_out = nullptr; p.cs_partial = nullptr; }   // column sums: separate pass below
    if (kind == TN_SMALL) hipLaunchKernelGGL(gemm_bf16_tn_kernel, dim3(grid), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(gemm_bf16_tn_big_kernel, dim3(grid), dim3(512), 0, stream, p);
    SC_CHECK_LAUNCH();
    if (p.splits > 1) {
        const int64_t mn = m * n;
#define SKR(SL) hipLaunchKernelGGL(splitk_reduce_kernel<SL>, dim3((unsigned)sc_cdiv(mn / 4, 256 / SL)), dim3(256), 0, stream, p.partial, p.splits, mn, (int)n, \
                                   c, ldc, alpha, beta, p.cs_partial, colsum_a, (int)m, colsum_beta)
        if (p.splits >= 64) SKR(16);
        else if (p.splits >= 16) SKR(4);
        else SKR(1);
#undef SKR
        SC_CHECK_LAUNCH();
    }
    if (kind == TN_BIG && colsum_a)   // stream-ordered behind the reduction, so the partial-slab workspace is free again
        return sc_colsum(a, SC_BF16, r, m, lda, colsum_a, colsum_beta != 0.f ? 1 : 0, ws, ws_bytes, (void*)stream);
    return SC_OK;
}

// ws bytes of a grouped weight-gradient launch
size_t sc_gemm_bf16_tn_group_ws(int nprob, const int64_t* m, const int64_t* n, int64_t r) {
    int64_t tiles = 0, elems = 0;
    for (int k = 0; k < nprob; ++k) { tiles += sc_cdiv(m[k], B_M) * sc_cdiv(n[k], B_N); elems += m[k] * n[k]; }
    int64_t s = tiles > 0 ? sc_num_cus() / tiles : 1;
    const int64_t nk = r / KSTEP, cap = nk / 4 > 1 ? nk / 4 : 1;
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    return s > 1 ? (size_t)s * elems * sizeof(float) : 0;
}

// dW_k (+)= A_k^T B_k for up to four problems sharing the contraction length r (r % 64 == 0, every M_k, N_k a multiple of 8)
int sc_gemm_bf16_tn_group_launch(int nprob, const int64_t* m, const int64_t* n, int64_t r, const void* const* a, const int64_t* lda, const void* const* b,
                                 const int64_t* ldb, float* const* c, const int64_t* ldc, float alpha, float beta, void* ws, size_t ws_bytes,
                                 hipStream_t stream) {
    SC_REQUIRE(nprob >= 1 && nprob <= 4 && r > 0 && r % KSTEP == 0, SC_ERR_SHAPE, "sc_gemm_bf16_tn_group: 1..4 problems, r %% 64 == 0");
    TnGroup g;
    g.n = nprob; g.K = (int)r; g.alpha = alpha; g.beta = beta;
    int64_t tiles = 0, elems = 0;
    for (int k = 0; k < nprob; ++k) {
        SC_REQUIRE(m[k] > 0 && n[k] > 0 && m[k] % 8 == 0 && n[k] % 8 == 0 && a[k] && b[k] && c[k], SC_ERR_SHAPE, "sc_gemm_bf16_tn_group: bad problem %d", k);
        SC_REQUIRE(lda[k] % 8 == 0 && ldb[k] % 8 == 0 && ldc[k] % 4 == 0 && lda[k] >= m[k] && ldb[k] >= n[k] && ldc[k] >= n[k], SC_ERR_SHAPE,
                   "sc_gemm_bf16_tn_group: bad leading dimension (problem %d)", k);
        SC_REQUIRE(sc_aligned(a[k], 16) && sc_aligned(b[k], 16) && sc_aligned(c[k], 16), SC_ERR_ALIGN, "sc_gemm_bf16_tn_group: operands must be 16-byte aligned");
        TnProb& q = g.prob[k];
        q.A = (const bf16_t*)a[k]; q.B = (const bf16_t*)b[k]; q.C = c[k]; q.M = (int)m[k]; q.N = (int)n[k];
        q.lda = lda[k]; q.ldb = ldb[k]; q.ldc = ldc[k];
        q.tiles_m = (int)sc_cdiv(m[k], B_M); q.tiles_n = (int)sc_cdiv(n[k], B_N);
        tiles += (int64_t)q.tiles_m * q.tiles_n; elems += m[k] * n[k];
    }
    for (int k = nprob; k < 4; ++k) g.prob[k] = g.prob[0];
    const int64_t nk = r / KSTEP, cap = nk / 4 > 1 ? nk / 4 : 1;
    int64_t s = sc_num_cus() / tiles;
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    g.k_per_split = (int)sc_cdiv(nk, s);
    g.splits = (int)sc_cdiv(nk, g.k_per_split);
    float* slab = (float*)ws;
    if (g.splits > 1) {
        SC_REQUIRE(ws && sc_aligned(ws, 16) && ws_bytes >= (size_t)g.splits * elems * sizeof(float), SC_ERR_WORKSPACE, "sc_gemm_bf16_tn_group: workspace too small");
    }
    int first = 0;
    int fb[4] = {0, 0, 0, 0};
    int blocks = 0;
    for (int k = 0; k < nprob; ++k) {
        TnProb& q = g.prob[k];
        q.first = first;
        first += q.tiles_m * q.tiles_n * g.splits;
        q.partial = g.splits > 1 ? slab : nullptr;
        slab += (size_t)g.splits * m[k] * n[k];
        fb[k] = blocks;
        blocks += (int)sc_cdiv(m[k] * n[k] / 4, 256);
    }
    for (int k = nprob; k < 4; ++k) fb[k] = blocks;
    g.total = first;
    hipLaunchKernelGGL(gemm_bf16_tn_group_kernel, dim3((unsigned)g.total), dim3(512), 0, stream, g);
    SC_CHECK_LAUNCH();
    if (g.splits > 1) {
        hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, g, int4{fb[0], fb[1], fb[2], fb[3]});
        SC_CHECK_LAUNCH();
    }
    return SC_OK;
}

extern "C" int sc_gemm_bf16_tn_group(int nprob, const int64_t* m, const int64_t* n, int64_t r, const void* const* a, const int64_t* lda,
                                     const void* const* b, const int64_t* ldb, float* const* c, const int64_t* ldc, float alpha, float beta, void* ws,
                                     size_t ws_bytes, void* stream) {
    SC_REQUIRE(m && n && a && lda && b && ldb && c && ldc, SC_ERR_ARG, "sc_gemm_bf16_tn_group: null argument");
    return sc_gemm_bf16_tn_group_launch(nprob, m, n, r, a, lda, b, ldb, c, ldc, alpha, beta, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" size_t sc_gemm_bf16_tn_group_workspace_bytes(int nprob, const int64_t* m, const int64_t* n, int64_t r) {
    if (!m || !n || nprob < 1 || nprob > 4 || r <= 0) return 0;
    return sc_gemm_bf16_tn_group_ws(nprob, m, n, r);
}

extern "C" int sc_gemm_bf16_nt(int64_t m, int64_t n, int64_t k, const void* a, int64_t lda, const void* b, int64_t ldb, void* c, int64_t ldc,
                               int out_dtype, const sc_gemm_epilogue* e, void* stream) {
    EpiParams epi;
    SC_TRY(epi_from_abi(e, SC_BF16, epi));
    return sc_gemm_bf16_nt_launch(m, n, k, a, lda, b, ldb, c, ldc, out_dtype, epi, (hipStream_t)stream);
}
// Diagnostic hook, deliberately NOT in include/sparsify_hip.h: tools/gemm_stamps.py sets a device buffer ([tiles][4] uint64) that the
// stamped instances of the persistent NT kernel fill with s_memtime values; NULL (the default) selects the production instances.
extern "C" void sc_gemm_bf16_nt_stamps(void* buf) { g_nt_stamps = (unsigned long long*)buf; }

// Diagnostic hook (not in the header): `blocks` workgroups that each keep a CU's registers and LDS for `microseconds` - a stand-in for
// a collective's kernels when tools/corun_bench.py measures how a GEMM behaves with part of the GPU taken.
namespace {
__global__ __launch_bounds__(512) void occupy_kernel(long long ticks, float* sink) {
    __shared__ float hold[16384];   // 64 KiB: with the persistent GEMM's 160 KiB no GEMM workgroup fits beside this one
    hold[threadIdx.x] = (float)threadIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (ticks < 0) sink[0] = hold[(threadIdx.x * 7) & 16383];
}
}  // namespace
// Diagnostic hook (not in the header): leaves every CU's LDS full of NaN bit patterns, so that a test of a kernel that reads LDS it
// has not written fails every time instead of once in a while (LDS is not cleared between kernels).
namespace {
__global__ __launch_bounds__(256) void poison_lds_kernel(float* sink) {
    extern __shared__ unsigned poison[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) poison[i] = 0xffffffffu;
    __syncthreads();
    if (sink) sink[0] = __uint_as_float(poison[threadIdx.x]);
}
}  // namespace
extern "C" int sc_debug_poison_lds(void* stream) {
    hipError_t e = hipFuncSetAttribute((const void*)poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return sc_set_error((int)e, "sc_debug_poison_lds: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(poison_lds_kernel, dim3(1024), dim3(256), 160 * 1024, (hipStream_t)stream, (float*)nullptr);
    SC_CHECK_LAUNCH();
    return SC_OK;
}
extern "C" int sc_debug_occupy(int blocks, int microseconds, void* stream) {
    hipLaunchKernelGGL(occupy_kernel, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, (long long)microseconds * 100, (float*)nullptr);
    SC_CHECK_LAUNCH();
    return SC_OK;
}

// 3x3 convolution, stride 1, padding 1, as an implicit GEMM on the 256x128 kernel (see GemmBf16Params::conv_w): no patch matrix.
// a_halo: NHWC bf16 [batch][h + 2][w + 2][cin] with a zero one-pixel border; b: [n][9 * cin] (tap-major, cin % 64 == 0, cin a power-of-two
// multiple of 64); c: [batch * h * w][n] bf16 or fp32, compact rows.
extern "C" int sc_conv3x3_bf16(const void* a_halo, const void* b, void* c, int out_dtype, int64_t batch, int64_t h, int64_t w, int64_t cin, int64_t n,
                               const sc_gemm_epilogue* e, void* stream) {
    SC_REQUIRE(a_halo && b && c && batch > 0 && h > 0 && w > 0, SC_ERR_ARG, "sc_conv3x3_bf16: bad argument");
    const int64_t kpt = cin / KSTEP;
    SC_REQUIRE(cin % KSTEP == 0 && kpt >= 1 && (kpt & (kpt - 1)) == 0, SC_ERR_SHAPE, "sc_conv3x3_bf16: cin = %lld must be 64 times a power of two", (long long)cin);
    SC_REQUIRE(n % 8 == 0 && batch * h * w < (1ll << 31), SC_ERR_SHAPE, "sc_conv3x3_bf16: bad sizes");
    SC_REQUIRE(out_dtype == SC_BF16 || out_dtype == SC_F32, SC_ERR_DTYPE, "sc_conv3x3_bf16: bad out dtype");
    SC_REQUIRE(sc_aligned(a_halo, 16) && sc_aligned(b, 16) && sc_aligned(c, 16), SC_ERR_ALIGN, "sc_conv3x3_bf16: operands must be 16-byte aligned");
    EpiParams epi;
    SC_TRY(epi_from_abi(e, SC_BF16, epi));
    SC_REQUIRE(!epi.colsum && !epi.pre_out && !epi.dgelu_pre && epi.act == 0, SC_ERR_ARG, "sc_conv3x3_bf16: only bias / residual epilogues");
    SC_REQUIRE(!epi.resid || epi.ld_aux % 8 == 0, SC_ERR_SHAPE, "sc_conv3x3_bf16: ld_aux must be a multiple of 8");
    GemmBf16Params p;
    p.A = (const bf16_t*)a_halo; p.B = (const bf16_t*)b; p.C = c;
    p.M = (int)(batch * h * w); p.N = (int)n; p.K = (int)(9 * cin);
    p.lda = cin; p.ldb = 9 * cin; p.ldc = n;
    p.splits = 1; p.k_per_split = 0; p.partial = nullptr;
    p.stamps = nullptr;
    p.epi = epi;
    p.epi.cs_partial = nullptr; p.epi.tickets = nullptr;
    p.conv_w = (int)w; p.conv_h = (int)h;
    p.conv_kpt_log2 = 0;
    while ((1ll << p.conv_kpt_log2) < kpt) ++p.conv_kpt_log2;
    p.tiles_m = (int)sc_cdiv(p.M, T_M); p.tiles_n = (int)sc_cdiv(n, T_N);
    p.group_m = GROUP_M; p.group_n = GROUP_N; p.half_tiles = 0; p.half_m0 = 0;
    const unsigned grid = (unsigned)(p.tiles_m * p.tiles_n);
    if (out_dtype == SC_F32) hipLaunchKernelGGL(gemm_bf16_nt256_kernel<true>, dim3(grid), dim3(512), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(gemm_bf16_nt256_kernel<false>, dim3(grid), dim3(512), 0, (hipStream_t)stream, p);
    SC_CHECK_LAUNCH();
    return SC_OK;
}

// Weight gradient of sc_conv3x3_bf16: dw[cout][tap * cin + c] = alpha * sum over bordered rows r of dz[r][cout] * x[r + shift(tap)][c] + beta * dw.
// dz_halo / x_halo: bordered NHWC images [batch][h + 2][w + 2][.]; dz's border is zero (so border rows add nothing) and x_halo has
// w + 3 rows of finite values (zeros) readable before and after the image.  Workspace: sc_gemm_bf16_tn_workspace_bytes(cout, 9 cin, rows).
extern "C" int sc_conv3x3_dw_bf16(const void* dz_halo, const void* x_halo, float* dw, int64_t batch, int64_t h, int64_t w, int64_t cout, int64_t cin,
                                  float alpha, float beta, void* ws, size_t ws_bytes, void* stream) {
    SC_REQUIRE(batch > 0 && h > 0 && w > 0 && w < 32768 && cin > 0 && cin % 8 == 0, SC_ERR_SHAPE, "sc_conv3x3_dw_bf16: bad sizes");
    SC_REQUIRE((w + 3) * cin * 4 + 64 * cin * 2 < (1ll << 31), SC_ERR_SHAPE, "sc_conv3x3_dw_bf16: row shift out of the 32-bit offset range");
    return sc_gemm_bf16_tn_launch(cout, 9 * cin, batch * (h + 2) * (w + 2), dz_halo, cout, x_halo, cin, dw, 9 * cin, alpha, beta, ws, ws_bytes,
                                  (hipStream_t)stream, nullptr, 0.f, (int)w);
}

extern "C" size_t sc_gemm_bf16_tn_workspace_bytes(int64_t m, int64_t n, int64_t r) {
    if (m <= 0 || n <= 0 || r <= 0) return 0;
    return sc_gemm_bf16_tn_ws(m, n, r);
}
extern "C" int sc_gemm_bf16_tn(int64_t m, int64_t n, int64_t r, const void* a, int64_t lda, const void* b, int64_t ldb, float* c, int64_t ldc,
                               float alpha, float beta, void* ws, size_t ws_bytes, void* stream) {
    return sc_gemm_bf16_tn_launch(m, n, r, a, lda, b, ldb, c, ldc, alpha, beta, ws, ws_bytes, (hipStream_t)stream, nullptr, 0.f, 0);
}
extern "C" int sc_gemm_bf16_tn_colsum(int64_t m, int64_t n, int64_t r, const void* a, int64_t lda, const void* b, int64_t ldb, float* c, int64_t ldc,
                                      float alpha, float beta, float* colsum_a, float colsum_beta, void* ws, size_t ws_bytes, void* stream) {
    SC_REQUIRE(colsum_a != nullptr, SC_ERR_ARG, "sc_gemm_bf16_tn_colsum: null colsum_a");
    return sc_gemm_bf16_tn_launch(m, n, r, a, lda, b, ldb, c, ldc, alpha, beta, ws, ws_bytes, (hipStream_t)stream, colsum_a, colsum_beta, 0);
}
