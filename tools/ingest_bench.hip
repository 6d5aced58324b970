// Micro-benchmark: how fast can one CU pull GEMM operand tiles into LDS with global_load_lds, as a function of the
// bytes kept in flight?  Same tile walk (XCD remap + band/cell order) and the same staging pattern as the NT GEMM
// kernels, but no MFMAs and (optionally) no fragment reads.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ingest tools/ingest_bench.hip && /tmp/ingest
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned short bf16_t;

__device__ __forceinline__ int xcd_remap(int id, int nwg) {
    const int xcd = id & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
}

// TM x TN tile, K-step KB bytes per row (128 = 64 bf16), S stages, W waves.  Each stage = (TM + TN) rows x KB bytes.
template <int TM, int TN, int KB, int S, int W, int NRD>
__global__ __launch_bounds__(W * 64) void ingest_kernel(const bf16_t* A, const bf16_t* B, int M, int N, int K, int tiles_m, int tiles_n,
                                                          float* sink) {
    constexpr int STAGE = (TM + TN) * KB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr int GM = 8, GN = 4;
    const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int band = tile / (GM * tiles_n), r_band = tile - band * (GM * tiles_n);
    const int rows = min(GM, tiles_m - band * GM);
    const int cell = r_band / (rows * GN), r_cell = r_band - cell * (rows * GN);
    const int gw = min(GN, tiles_n - cell * GN);
    const int m0 = (band * GM + r_cell / gw) * TM, n0 = (cell * GN + r_cell % gw) * TN;
    constexpr int LPR = KB / 16;             // lanes per row
    constexpr int RPI = 64 / LPR;            // rows per wave instruction
    constexpr int NI = (TM + TN) / RPI / W;  // instructions per wave per stage
    const int srow = lane / LPR, schunk = lane % LPR;
    const char* gp[NI];
#pragma unroll
    for (int q = 0; q < NI; ++q) {
        const int r = (wave * NI + q) * RPI + srow;   // row in the stacked [A rows; B rows] tile
        gp[q] = r < TM ? (const char*)(A + (size_t)min(m0 + r, M - 1) * K) + schunk * 16 : (const char*)(B + (size_t)min(n0 + r - TM, N - 1) * K) + schunk * 16;
    }
    const int nk = K * 2 / KB;
    float acc = 0.f;
    auto stage = [&](int s, int kt) {
#pragma unroll
        for (int q = 0; q < NI; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(gp[q] + (size_t)kt * KB), (lptr_t)(smem + s * STAGE + (wave * NI + q) * 1024), 16, 0, 0);
    };
    for (int s = 0; s < S - 1 && s < nk; ++s) stage(s, s);
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt must have landed: at most min(S-2, nk-1-kt) younger tiles may stay in flight
        const int younger = min(S - 2, nk - 1 - kt);
        if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NI) : "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + S - 1 < nk) stage((kt + S - 1) % S, kt + S - 1);
        if (NRD > 0) {   // NRD 1-KiB fragment reads (ds_read_b128) per wave per K-tile, as the MFMA loop would issue them
            const char* base = smem + (kt % S) * STAGE;
            constexpr int NFR = (TM + TN) * KB / 1024;
#pragma unroll
            for (int i = 0; i < NRD; ++i) {
                const float4 v = *(const float4*)(base + ((i * W + wave) % NFR) * 1024 + lane * 16);
                acc += v.x + v.w;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int TM, int TN, int KB, int S, int W, int NRD>
void run(const char* name, const bf16_t* A, const bf16_t* B, int M, int N, int K, float* sink) {
    const int tm = (M + TM - 1) / TM, tn = (N + TN - 1) / TN;
    const size_t lds = (size_t)S * (TM + TN) * KB;
    auto kern = ingest_kernel<TM, TN, KB, S, W, NRD>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(tm * tn), dim3(W * 64), lds, 0, A, B, M, N, K, tm, tn, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)tm * tn * (TM + TN) * K * 2.0;
    hipError_t err = hipGetLastError();
    printf("%-44s M=%d N=%d K=%d  lds=%3zu KB  %8.1f us  %7.1f GB/s/CU  (%.2f TB/s)  flop-equiv %.0f TF %s\n", name, M, N, K, lds / 1024, best * 1e3,
           bytes / (best * 1e-3) / 256 / 1e9, bytes / (best * 1e-3) / 1e12, 2.0 * M * N * K / (best * 1e-3) / 1e12, err ? hipGetErrorString(err) : "");
    fflush(stdout);
}

// LDS read peak: W waves per workgroup (one workgroup per CU, 2 when W == 16 is split), every wave streams 1-KiB lane-linear fragments.
// WIDTH = bytes per lane per instruction (4 / 8 / 16); TR = ds_read_b64_tr_b16.
template <int WIDTH, bool TR>
__global__ __launch_bounds__(1024) void lds_read_kernel(float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 16384; i += blockDim.x) ((float*)smem)[i] = (float)i;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(lptr_t)smem + ((wave * 2048) & 32767) + lane * WIDTH;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            if (TR) {
                float2 v; asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(f * 512)); 
                asm volatile("" : "+v"(v)); acc += 0.f;
            } else if (WIDTH == 16) {
                float4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(f * 1024));
            } else if (WIDTH == 8) {
                float2 v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(f * 512));
            } else {
                float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(f * 256));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int WIDTH, bool TR>
void run_lds(const char* name, int waves, float* sink) {
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((lds_read_kernel<WIDTH, TR>), dim3(256), dim3(waves * 64), 0, 0, sink, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)iters * 16 * 64 * WIDTH * waves;   // per CU
    printf("LDS read %-28s waves/CU=%2d  %8.1f us  %7.1f GB/s/CU  = %5.1f B/clk @2.4GHz, %5.1f B/clk @2.1GHz\n", name, waves, best * 1e3,
           bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 2.4e9, bytes / (best * 1e-3) / 2.1e9);
    fflush(stdout);
}

// The NT main loop taken apart: LDS-DMA ring as above + (optionally) NRD asm fragment reads and NMF MFMAs per wave per K-tile,
// in the real kernels' order (sub-step 0 MFMAs | barrier | stage | sub-step 1 MFMAs), inline assembly so the compiler adds no waits.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
template <int TM, int TN, int S, int W, int NRD, int NMF, bool STORE>
__global__ __launch_bounds__(W * 64) void loop_kernel(const bf16_t* A, const bf16_t* B, bf16_t* C, int M, int N, int K, int tiles_m, int tiles_n, float* sink) {
    constexpr int KB = 128, STAGE = (TM + TN) * KB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr int GM = 8, GN = 4;
    const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int band = tile / (GM * tiles_n), r_band = tile - band * (GM * tiles_n);
    const int rows = min(GM, tiles_m - band * GM);
    const int cell = r_band / (rows * GN), r_cell = r_band - cell * (rows * GN);
    const int gw = min(GN, tiles_n - cell * GN);
    const int m0 = (band * GM + r_cell / gw) * TM, n0 = (cell * GN + r_cell % gw) * TN;
    constexpr int NI = (TM + TN) / 8 / W;
    const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
    const char* gp[NI];
#pragma unroll
    for (int q = 0; q < NI; ++q) {
        const int r = (wave * NI + q) * 8 + srow;
        gp[q] = r < TM ? (const char*)(A + (size_t)min(m0 + r, M - 1) * K) + schunk * 16 : (const char*)(B + (size_t)min(n0 + r - TM, N - 1) * K) + schunk * 16;
    }
    const int nk = K / 64;
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    s16x8 fr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) fr[i] = s16x8{1, 2, 3, 4, 5, 6, 7, 8};
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const unsigned rbase = lds0 + ((wave * 8 + (lane & 15)) * 128) + (((lane >> 4) ^ (lane & 7)) * 16);   // conflict-free swizzled fragment address
    auto stage = [&](int s, int kt) {
#pragma unroll
        for (int q = 0; q < NI; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(gp[q] + (size_t)kt * KB), (lptr_t)(smem + s * STAGE + (wave * NI + q) * 1024), 16, 0, 0);
    };
    for (int s = 0; s < S - 1 && s < nk; ++s) stage(s, s);
    for (int kt = 0; kt < nk; ++kt) {
        const unsigned rb = rbase + (kt % S) * STAGE;
        // sub-step 0: half of the MFMAs with half of the reads woven in
#pragma unroll
        for (int i = 0; i < NMF / 2; ++i) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i % 16]) : "v"(fr[i % 4]), "v"(fr[4 + i % 4]));
            if (NRD > 0 && i % (NMF / NRD) == 0 && i / (NMF / NRD) < NRD / 2)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(i / (NMF / NRD)) % 8]) : "v"(rb), "n"((i / (NMF / NRD)) * 2048) : "memory");
        }
        const int younger = min(S - 2, nk - 1 - kt);
        if (younger >= 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + S - 1 < nk) stage((kt + S - 1) % S, kt + S - 1);
#pragma unroll
        for (int i = 0; i < NMF / 2; ++i) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i % 16]) : "v"(fr[i % 4]), "v"(fr[4 + i % 4]));
            if (NRD > 0 && i % (NMF / NRD) == 0 && i / (NMF / NRD) < NRD / 2)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(i / (NMF / NRD)) % 8]) : "v"(rb), "n"((i / (NMF / NRD)) * 2048 + 64) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += acc[i][0] + acc[i][3];
    if (STORE) {   // a bf16 C tile's worth of coalesced 16-byte stores
        constexpr int PER_LANE = TM * TN * 2 / 16 / (W * 64);
        char* cp = (char*)C + ((size_t)(m0) * N + n0) * 2;
#pragma unroll 4
        for (int i = 0; i < PER_LANE; ++i) {
            const int idx = i * W * 64 + t;               // 16-byte piece of the tile, row-major, TN*2/16 pieces per row
            const int r = idx / (TN / 8), c = idx % (TN / 8);
            if (m0 + r < M && n0 + c * 8 < N) *(float4*)(cp + ((size_t)r * N + c * 8) * 2) = float4{tot, tot, tot, tot};
        }
    } else if (tot == 123.456f) sink[0] = tot;
}

template <int TM, int TN, int S, int W, int NRD, int NMF, bool STORE>
void run_loop(const char* name, const bf16_t* A, const bf16_t* B, bf16_t* C, int M, int N, int K, float* sink) {
    const int tm = (M + TM - 1) / TM, tn = (N + TN - 1) / TN;
    const size_t lds = (size_t)S * (TM + TN) * 128;
    auto kern = loop_kernel<TM, TN, S, W, NRD, NMF, STORE>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(tm * tn), dim3(W * 64), lds, 0, A, B, C, M, N, K, tm, tn, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    hipError_t err = hipGetLastError();
    printf("loop %-50s M=%d N=%d K=%d  %8.1f us  flop-equiv %6.0f TF %s\n", name, M, N, K, best * 1e3, 2.0 * M * N * K / (best * 1e-3) / 1e12,
           err ? hipGetErrorString(err) : "");
    fflush(stdout);
}

// 256x256 tile, 8 waves, the whole 160 KiB of LDS as a ring of five 32-KiB half-stages (the A rows or the B rows of one K-tile):
// A(t) -> slot 2t % 5, B(t) -> slot (2t+1) % 5.  At the mid-tile barrier of tile t the slots of A(t), B(t) are free and take
// B(t+2), A(t+3); A(t+2) (requested one barrier earlier) is still in flight, so 96 KiB are requested ahead instead of 64.
template <int NRD, int NMF, bool STORE>
__global__ __launch_bounds__(512) void ring_kernel(const bf16_t* A, const bf16_t* B, bf16_t* C, int M, int N, int K, int tiles_m, int tiles_n, float* sink) {
    constexpr int TM = 256, TN = 256, SLOT = 256 * 128, W = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr int GM = 8, GN = 4;
    const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int band = tile / (GM * tiles_n), r_band = tile - band * (GM * tiles_n);
    const int rows = min(GM, tiles_m - band * GM);
    const int cell = r_band / (rows * GN), r_cell = r_band - cell * (rows * GN);
    const int gw = min(GN, tiles_n - cell * GN);
    const int m0 = (band * GM + r_cell / gw) * TM, n0 = (cell * GN + r_cell % gw) * TN;
    const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
    const char *ga[4], *gb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = (wave * 4 + q) * 8 + srow;
        ga[q] = (const char*)(A + (size_t)min(m0 + r, M - 1) * K) + schunk * 16;
        gb[q] = (const char*)(B + (size_t)min(n0 + r, N - 1) * K) + schunk * 16;
    }
    const int nk = K / 64;
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    s16x8 fr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) fr[i] = s16x8{1, 2, 3, 4, 5, 6, 7, 8};
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const unsigned rbase = lds0 + ((wave * 8 + (lane & 15)) * 128) + (((lane >> 4) ^ (lane & 7)) * 16);
    auto stage = [&](const char* const (&g)[4], int slot, int kt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) __builtin_amdgcn_global_load_lds((gptr_t)(g[q] + (size_t)kt * 128), (lptr_t)(smem + slot * SLOT + (wave * 4 + q) * 1024), 16, 0, 0);
    };
    stage(ga, 0, 0); stage(gb, 1, 0);
    if (nk > 1) { stage(ga, 2, 1); stage(gb, 3, 1); }
    if (nk > 2) stage(ga, 4, 2);
    if (nk > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int sa = 0;   // slot of A(kt)
    for (int kt = 0; kt < nk; ++kt) {
        const int sb = sa + 1 >= 5 ? sa - 4 : sa + 1, sa1 = sa + 2 >= 5 ? sa - 3 : sa + 2, sb1 = sa + 3 >= 5 ? sa - 2 : sa + 3;
        const unsigned ra = rbase + sa * SLOT, rb = rbase + sb * SLOT, ra1 = rbase + sa1 * SLOT, rb1 = rbase + sb1 * SLOT;
#pragma unroll
        for (int i = 0; i < NMF / 2; ++i) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i % 16]) : "v"(fr[i % 4]), "v"(fr[4 + i % 4]));
            if (NRD > 0 && i % 4 == 0) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(i / 4) % 8]) : "v"(ra), "n"((i / 4) * 2048 + 64) : "memory");
                if (i / 4 < 4) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(i / 4 + 4) % 8]) : "v"(rb), "n"((i / 4) * 2048 + 64) : "memory");
            }
        }
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) stage(gb, sa, kt + 2);
        if (kt + 3 < nk) stage(ga, sb, kt + 3);
#pragma unroll
        for (int i = 0; i < NMF / 2; ++i) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i % 16]) : "v"(fr[i % 4]), "v"(fr[4 + i % 4]));
            if (NRD > 0 && i % 4 == 0) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(i / 4) % 8]) : "v"(ra1), "n"((i / 4) * 2048) : "memory");
                if (i / 4 < 4) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(i / 4 + 4) % 8]) : "v"(rb1), "n"((i / 4) * 2048) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        sa = sa1;
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += acc[i][0] + acc[i][3];
    if (STORE) {
        constexpr int PER_LANE = TM * TN * 2 / 16 / (W * 64);
        char* cp = (char*)C + ((size_t)(m0) * N + n0) * 2;
#pragma unroll 4
        for (int i = 0; i < PER_LANE; ++i) {
            const int idx = i * W * 64 + t;
            const int r = idx / (TN / 8), c = idx % (TN / 8);
            if (m0 + r < M && n0 + c * 8 < N) *(float4*)(cp + ((size_t)r * N + c * 8) * 2) = float4{tot, tot, tot, tot};
        }
    } else if (tot == 123.456f) sink[0] = tot;
}

template <int NRD, int NMF, bool STORE>
void run_ring(const char* name, const bf16_t* A, const bf16_t* B, bf16_t* C, int M, int N, int K, float* sink) {
    const int tm = (M + 255) / 256, tn = (N + 255) / 256;
    const size_t lds = 5 * 256 * 128;
    auto kern = ring_kernel<NRD, NMF, STORE>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(tm * tn), dim3(512), lds, 0, A, B, C, M, N, K, tm, tn, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    hipError_t err = hipGetLastError();
    printf("ring %-50s M=%d N=%d K=%d  %8.1f us  flop-equiv %6.0f TF %s\n", name, M, N, K, best * 1e3, 2.0 * M * N * K / (best * 1e-3) / 1e12,
           err ? hipGetErrorString(err) : "");
    fflush(stdout);
}

int main() {
    if (getenv("RING_ONLY")) {
        for (int pass = 0; pass < 2; ++pass) {
            const int M = 51200, N = pass ? 768 : 3072, K = pass ? 3072 : 768;
            bf16_t *A, *B, *C; float* sk;
            hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&sk, 64);
            std::vector<bf16_t> h((size_t)M * K);
            unsigned x = 12345u;
            for (auto& v : h) { x = x * 1664525u + 1013904223u; const float f = ((x >> 8) & 0xffff) / 32768.f - 1.f + ((x >> 4) & 0xff) / 512.f; unsigned u; memcpy(&u, &f, 4); v = (bf16_t)(u >> 16); }
            hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice);
            hipMemcpy(B, h.data() + 777, (size_t)N * K * 2, hipMemcpyHostToDevice);
            run_loop<256, 256, 2, 8, 0, 0, false>("256x256 S2 W8 loads only", A, B, C, M, N, K, sk);
            run_ring<0, 0, false>("256x256 ring5 W8 loads only", A, B, C, M, N, K, sk);
            run_loop<256, 256, 2, 8, 0, 64, false>("256x256 S2 W8 loads+64mfma", A, B, C, M, N, K, sk);
            run_ring<0, 64, false>("256x256 ring5 W8 loads+64mfma", A, B, C, M, N, K, sk);
            run_loop<256, 256, 2, 8, 24, 64, false>("256x256 S2 W8 full loop", A, B, C, M, N, K, sk);
            run_ring<24, 64, false>("256x256 ring5 W8 full loop", A, B, C, M, N, K, sk);
            run_loop<256, 256, 2, 8, 24, 64, true>("256x256 S2 W8 full loop + C stores", A, B, C, M, N, K, sk);
            run_ring<24, 64, true>("256x256 ring5 W8 full loop + C stores", A, B, C, M, N, K, sk);
            hipFree(A); hipFree(B); hipFree(C); hipFree(sk);
        }
        return 0;
    }
    if (getenv("LOOP_ONLY")) {
        const int M = 51200, N = 3072, K = 768;
        bf16_t *A, *B, *C; float* sk;
        hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&sk, 64);
        hipMemset(A, 0, (size_t)M * K * 2); hipMemset(B, 0, (size_t)N * K * 2);
        if (getenv("RANDOM_DATA")) {   // N(0,1)-like bf16 operands: MFMA power (and with it the clock) depends on the data
            std::vector<bf16_t> h((size_t)M * K);
            unsigned x = 12345u;
            for (auto& v : h) { x = x * 1664525u + 1013904223u; const float f = ((x >> 8) & 0xffff) / 32768.f - 1.f + ((x >> 4) & 0xff) / 512.f; unsigned u; memcpy(&u, &f, 4); v = (bf16_t)(u >> 16); }
            hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice);
            hipMemcpy(B, h.data() + 777, (size_t)N * K * 2, hipMemcpyHostToDevice);
            printf("random operands\n");
        }
        run_loop<256, 128, 3, 8, 16, 32, false>("256x128 S3 W8 loads+16rd+32mfma (full loop)", A, B, C, M, N, K, sk);
        run_loop<256, 128, 3, 8, 16, 32, true>("256x128 S3 W8 full loop + C stores", A, B, C, M, N, K, sk);
        run_loop<256, 128, 3, 8, 0, 32, false>("256x128 S3 W8 loads+32mfma", A, B, C, M, N, K, sk);
        run_loop<128, 128, 2, 4, 16, 32, false>("128x128 S2 W4 full loop (2 WG/CU)", A, B, C, M, N, K, sk);
        run_loop<128, 128, 2, 4, 16, 32, true>("128x128 S2 W4 full loop + C stores", A, B, C, M, N, K, sk);
        run_loop<256, 128, 3, 8, 0, 0, false>("256x128 S3 W8 loads", A, B, C, M, N, K, sk);
        run_loop<256, 128, 3, 8, 16, 64, false>("256x128 S3 W8 loads+16rd+64mfma (full loop)", A, B, C, M, N, K, sk);
        run_loop<256, 128, 3, 8, 0, 64, false>("256x128 S3 W8 loads+64mfma", A, B, C, M, N, K, sk);
        run_loop<256, 128, 3, 8, 16, 64, true>("256x128 S3 W8 full loop + C stores", A, B, C, M, N, K, sk);
        run_loop<256, 128, 2, 8, 16, 64, false>("256x128 S2 W8 full loop", A, B, C, M, N, K, sk);
        run_loop<256, 256, 2, 4, 0, 0, false>("256x256 S2 W4 loads", A, B, C, M, N, K, sk);
        run_loop<256, 256, 2, 4, 32, 128, false>("256x256 S2 W4 loads+32rd+128mfma (full loop)", A, B, C, M, N, K, sk);
        run_loop<256, 256, 2, 4, 0, 128, false>("256x256 S2 W4 loads+128mfma", A, B, C, M, N, K, sk);
        run_loop<256, 256, 2, 4, 32, 128, true>("256x256 S2 W4 full loop + C stores", A, B, C, M, N, K, sk);
        run_loop<256, 256, 2, 8, 24, 64, false>("256x256 S2 W8 loads+24rd+64mfma (full loop)", A, B, C, M, N, K, sk);
        run_loop<256, 256, 2, 8, 24, 64, true>("256x256 S2 W8 full loop + C stores", A, B, C, M, N, K, sk);
        run_loop<128, 128, 2, 4, 16, 64, false>("128x128 S2 W4 full loop (2 WG/CU)", A, B, C, M, N, K, sk);
        run_loop<128, 128, 2, 4, 16, 64, true>("128x128 S2 W4 full loop + C stores", A, B, C, M, N, K, sk);
        return 0;
    }
    if (getenv("LDS_ONLY")) {
        float* sk; hipMalloc(&sk, 64);
        for (int w : {4, 8, 16}) { run_lds<16, false>("ds_read_b128", w, sk); run_lds<8, false>("ds_read_b64", w, sk); run_lds<4, false>("ds_read_b32", w, sk); run_lds<8, true>("ds_read_b64_tr_b16", w, sk); }
        return 0;
    }
    const int M = 51200, N = 3072, K = 768;
    bf16_t *A, *B; float* sink;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&sink, 64);
    hipMemset(A, 0x11, (size_t)M * K * 2); hipMemset(B, 0x22, (size_t)N * K * 2);
    run<256, 128, 128, 3, 8, 0>("256x128 BK64 S3 W8 (NT256 pattern)", A, B, M, N, K, sink);
    run<256, 128, 128, 2, 8, 0>("256x128 BK64 S2 W8", A, B, M, N, K, sink);
    run<256, 128, 64, 4, 8, 0>("256x128 BK32 S4 W8", A, B, M, N, K, sink);
    run<256, 128, 64, 6, 8, 0>("256x128 BK32 S6 W8", A, B, M, N, K, sink);
    run<256, 256, 128, 2, 4, 0>("256x256 BK64 S2 W4 (big pattern)", A, B, M, N, K, sink);
    run<256, 256, 128, 2, 8, 0>("256x256 BK64 S2 W8", A, B, M, N, K, sink);
    run<256, 256, 64, 4, 4, 0>("256x256 BK32 S4 W4", A, B, M, N, K, sink);
    run<256, 256, 64, 5, 4, 0>("256x256 BK32 S5 W4", A, B, M, N, K, sink);
    run<256, 256, 64, 4, 8, 0>("256x256 BK32 S4 W8", A, B, M, N, K, sink);
    run<128, 128, 128, 2, 4, 0>("128x128 BK64 S2 W4 (2 WG/CU)", A, B, M, N, K, sink);
    run<256, 128, 128, 3, 8, 16>("256x128 BK64 S3 W8 + 16 reads/wave", A, B, M, N, K, sink);
    run<256, 256, 128, 2, 4, 32>("256x256 BK64 S2 W4 + 32 reads/wave", A, B, M, N, K, sink);
    // long K: the steady state without the per-tile prologue
    const int K2 = 3072, N2 = 768;
    bf16_t *A2, *B2;
    hipMalloc(&A2, (size_t)M * K2 * 2); hipMalloc(&B2, (size_t)N2 * K2 * 2);
    hipMemset(A2, 0x11, (size_t)M * K2 * 2); hipMemset(B2, 0x22, (size_t)N2 * K2 * 2);
    run<256, 128, 128, 3, 8, 0>("256x128 BK64 S3 W8 K=3072", A2, B2, M, N2, K2, sink);
    run<256, 256, 128, 2, 4, 0>("256x256 BK64 S2 W4 K=3072", A2, B2, M, N2, K2, sink);
    run<256, 256, 64, 4, 4, 0>("256x256 BK32 S4 W4 K=3072", A2, B2, M, N2, K2, sink);
    return 0;
}
