// bf16 MFMA attention for the longer sequences (80 < S <= 272: ViT-L/14 has S = 257 image tokens).
// Same arithmetic and register choreography as attention_mfma.hip (scores of one query row on one lane, accumulator tile
// fed back as the next MFMA's B operand, transposed operands by ds_read_b64_tr_b16), but one WORKGROUP owns one
// (batch, head): the K / V (forward) or Q / K / dO (backward) images are shared in LDS by the 4 waves, which split the
// query tiles (and, in the backward's second pass, the key tiles) between them.  K and V fragments are read from LDS on
// demand instead of being kept in registers (17 key tiles do not fit).
#include "attention_mfma_common.h"
#include <stdlib.h>

namespace {

using namespace attn;

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

template <int NT, bool CAUSAL, int NW>
__global__ __launch_bounds__(64 * NW) void attn_fwd_long_kernel(const bf16_t* qkv, bf16_t* out, int S, int W, int H, float scale) {
    constexpr int KS = (NT + 1) / 2;
    constexpr bool ODD = (NT & 1) != 0;
    constexpr int IMG = NT * 16 * LDR;
    extern __shared__ __attribute__((aligned(16))) bf16_t lds_fl[];
    bf16_t* Ks = lds_fl;
    bf16_t* Vs = Ks + IMG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int64_t ld = 3 * (int64_t)W;
    const bf16_t* qb = qkv + (int64_t)b * S * ld + h * HD;
    stage_head_block(Ks, qb + W, ld, S, NT * 16, tid, 64 * NW);
    stage_head_block(Vs, qb + 2 * W, ld, S, NT * 16, tid, 64 * NW);
    __syncthreads();
    const int g = lane >> 4, c16 = lane & 15;
    const int n_it = (S + 15) >> 4;
    for (int it = wave; it < n_it; it += NW) {
        const bf16x8 q0 = row_frag_global(qb, ld, it, 0, lane, S), q1 = row_frag_global(qb, ld, it, 1, lane, S);
        const int i = it * 16 + c16;
        f32x4 sc[NT + 1];
        float m = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = MFMA16(row_frag_lds(Ks, jt, 0, lane), q0, a);
            a = MFMA16(row_frag_lds(Ks, jt, 1, lane), q1, a);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = jt * 16 + 4 * g + r;
                const bool ok = j < S && (!CAUSAL || j <= i);
                a[r] = ok ? a[r] * scale : -INFINITY;
                m = fmaxf(m, a[r]);
            }
            sc[jt] = a;
        }
        sc[NT] = f32x4{0.f, 0.f, 0.f, 0.f};
        m = group_max(m);
        float l = 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(sc[jt][r] - m);
                sc[jt][r] = p;
                l += p;
            }
        l = group_sum(l);
        const float inv = 1.0f / l;
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 pf = pack_frag(sc[2 * s], sc[2 * s + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 vt = (ODD && s == KS - 1) ? tr_frag<false>(Vs, s, 16 * dt, lane) : tr_frag<true>(Vs, s, 16 * dt, lane);
                o[dt] = MFMA16(vt, pf, o[dt]);
            }
        }
        if (i < S) {
            bf16_t* op = out + ((int64_t)b * S + i) * W + h * HD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) io<bf16_t>::st4(op + 16 * dt, o[dt] * inv);
        }
    }
}

template <int NT, bool CAUSAL, int NW>
__global__ __launch_bounds__(64 * NW, NT <= 8 ? 2 : 1) void attn_bwd_long_kernel(const bf16_t* qkv, const bf16_t* d_out, bf16_t* d_qkv, int S, int W, int H, float scale,
                                                                                      float* cs_part /* [batch][3 W] or null */) {
    constexpr int KS = (NT + 1) / 2;
    constexpr bool ODD = (NT & 1) != 0;
    constexpr int IMG = NT * 16 * LDR;
    extern __shared__ __attribute__((aligned(16))) bf16_t lds_bl[];
    bf16_t* Ks = lds_bl;
    bf16_t* Qs = Ks + IMG;
    bf16_t* Os = Qs + IMG;                  // dO
    float* st_m = (float*)(Os + IMG);
    float* st_il = st_m + NT * 16;
    float* st_dl = st_il + NT * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int64_t ld = 3 * (int64_t)W;
    const bf16_t* qb = qkv + (int64_t)b * S * ld + h * HD;
    const bf16_t* vb = qb + 2 * W;
    const bf16_t* dob = d_out + (int64_t)b * S * W + h * HD;
    bf16_t* dqb = d_qkv + (int64_t)b * S * ld + h * HD;
    stage_head_block(Qs, qb, ld, S, NT * 16, tid, 64 * NW);
    stage_head_block(Ks, qb + W, ld, S, NT * 16, tid, 64 * NW);
    stage_head_block(Os, dob, W, S, NT * 16, tid, 64 * NW);
    __syncthreads();
    const int g = lane >> 4, c16 = lane & 15;
    const int n_t = (S + 15) >> 4;

    f32x4 csq[4], csk[4], csv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) csq[dt] = csk[dt] = csv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---------------- pass 1: lane = query row i
    for (int it = wave; it < n_t; it += NW) {
        const bf16x8 q0 = row_frag_lds(Qs, it, 0, lane), q1 = row_frag_lds(Qs, it, 1, lane);
        const bf16x8 g0 = row_frag_lds(Os, it, 0, lane), g1 = row_frag_lds(Os, it, 1, lane);
        const int i = it * 16 + c16;
        f32x4 sc[NT + 1], dp[NT];
        float m = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            if (CAUSAL && jt > it) {   // key tile entirely in the future of this query tile: p = dS = 0, no work (wave-uniform)
                sc[jt] = dp[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
                continue;
            }
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};
            a = MFMA16(row_frag_lds(Ks, jt, 0, lane), q0, a);
            a = MFMA16(row_frag_lds(Ks, jt, 1, lane), q1, a);
            d = MFMA16(row_frag_global(vb, ld, jt, 0, lane, S), g0, d);
            d = MFMA16(row_frag_global(vb, ld, jt, 1, lane, S), g1, d);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = jt * 16 + 4 * g + r;
                const bool ok = j < S && (!CAUSAL || j <= i);
                a[r] = ok ? a[r] * scale : -INFINITY;
                m = fmaxf(m, a[r]);
            }
            sc[jt] = a;
            dp[jt] = d;
        }
        sc[NT] = f32x4{0.f, 0.f, 0.f, 0.f};
        m = group_max(m);
        float l = 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            if (CAUSAL && jt > it) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(sc[jt][r] - m);
                sc[jt][r] = p;
                l += p;
            }
        }
        l = group_sum(l);
        const float inv = 1.0f / l;
        float delta = 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            if (CAUSAL && jt > it) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sc[jt][r] *= inv;
                delta += sc[jt][r] * dp[jt][r];
            }
        }
        delta = group_sum(delta);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            if (CAUSAL && jt > it) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[jt][r] = sc[jt][r] * (dp[jt][r] - delta) * scale;
        }
        if (g == 0) {
            const bool live = i < S;
            st_m[i] = live ? m : 0.f;
            st_il[i] = live ? inv : 0.f;
            st_dl[i] = live ? delta : 0.f;
        }
        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (CAUSAL && 2 * s > it) continue;   // both key tiles of this k-step are masked out
            const bf16x8 dsf = pack_frag(sc[2 * s], sc[2 * s + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 kt = (ODD && s == KS - 1) ? tr_frag<false>(Ks, s, 16 * dt, lane) : tr_frag<true>(Ks, s, 16 * dt, lane);
                dq[dt] = MFMA16(kt, dsf, dq[dt]);
            }
        }
        if (i < S) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) io<bf16_t>::st4(dqb + (int64_t)i * ld + 16 * dt + 4 * g, dq[dt]);
        }
        if (cs_part) cs_add(csq, dq, i < S);
    }
    // tiles of padding queries (none when n_t == NT) must read as "no contribution" in pass 2
    for (int i = n_t * 16 + tid; i < NT * 16; i += 64 * NW) st_m[i] = st_il[i] = st_dl[i] = 0.f;
    __syncthreads();   // every query tile's statistics are in LDS

    // ---------------- pass 2: lane = key row j
    for (int jt = wave; jt < n_t; jt += NW) {
        const bf16x8 k0 = row_frag_lds(Ks, jt, 0, lane), k1 = row_frag_lds(Ks, jt, 1, lane);
        const bf16x8 v0 = row_frag_global(vb, ld, jt, 0, lane, S), v1 = row_frag_global(vb, ld, jt, 1, lane, S);
        const int j = jt * 16 + c16;
        f32x4 dv[4], dk[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dv[dt] = dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            f32x4 pt[2], dst[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                pt[u] = dst[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int it = 2 * s + u;
                if (ODD && it >= NT) continue;
                if (it >= n_t) continue;            // query tile beyond the sequence: pass 1 wrote no statistics for it
                if (CAUSAL && it < jt) continue;    // query tile entirely in the past of this key tile: P^T = dS^T = 0
                f32x4 a = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};
                a = MFMA16(row_frag_lds(Qs, it, 0, lane), k0, a);
                a = MFMA16(row_frag_lds(Qs, it, 1, lane), k1, a);
                d = MFMA16(row_frag_lds(Os, it, 0, lane), v0, d);
                d = MFMA16(row_frag_lds(Os, it, 1, lane), v1, d);
                const f32x4 mm = *(const f32x4*)(st_m + it * 16 + 4 * g), il = *(const f32x4*)(st_il + it * 16 + 4 * g),
                            dl = *(const f32x4*)(st_dl + it * 16 + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = it * 16 + 4 * g + r;
                    const bool ok = i < S && j < S && (!CAUSAL || j <= i);
                    const float p = ok ? __expf(a[r] * scale - mm[r]) * il[r] : 0.f;
                    pt[u][r] = p;
                    dst[u][r] = ok ? p * (d[r] - dl[r]) * scale : 0.f;   // not 0 * (d - dl): d, dl of a masked pair need not be finite
                }
            }
            if (CAUSAL && 2 * s + 1 < jt) continue;
            const bf16x8 pf = pack_frag(pt[0], pt[1]), dsf = pack_frag(dst[0], dst[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 ot = (ODD && s == KS - 1) ? tr_frag<false>(Os, s, 16 * dt, lane) : tr_frag<true>(Os, s, 16 * dt, lane);
                const bf16x8 qt = (ODD && s == KS - 1) ? tr_frag<false>(Qs, s, 16 * dt, lane) : tr_frag<true>(Qs, s, 16 * dt, lane);
                dv[dt] = MFMA16(ot, pf, dv[dt]);
                dk[dt] = MFMA16(qt, dsf, dk[dt]);
            }
        }
        if (j < S) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                io<bf16_t>::st4(dqb + (int64_t)j * ld + W + 16 * dt + 4 * g, dk[dt]);
                io<bf16_t>::st4(dqb + (int64_t)j * ld + 2 * W + 16 * dt + 4 * g, dv[dt]);
            }
        }
        if (cs_part) { cs_add(csk, dk, j < S); cs_add(csv, dv, j < S); }
    }
    if (cs_part) {   // per-wave sums over its tiles -> LDS -> the head's 192 columns for image b, waves added in a fixed order
        cs_rows(csq); cs_rows(csk); cs_rows(csv);
        __syncthreads();                       // the LDS images are dead
        float* red = (float*)lds_bl;           // [NW][192]
        if (c16 == 0) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                *(f32x4*)(red + wave * 192 + 16 * dt + 4 * g) = csq[dt];
                *(f32x4*)(red + wave * 192 + 64 + 16 * dt + 4 * g) = csk[dt];
                *(f32x4*)(red + wave * 192 + 128 + 16 * dt + 4 * g) = csv[dt];
            }
        }
        __syncthreads();
        if (tid < 192) {
            float v = red[tid];
#pragma unroll
            for (int w2 = 1; w2 < NW; ++w2) v += red[w2 * 192 + tid];
            cs_part[(int64_t)b * 3 * W + (tid >> 6) * W + h * HD + (tid & 63)] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward, long sequences
// Two sweeps over the key tiles per query tile (row max; then probabilities, row sum and P V per pair of key tiles, with the
// score product recomputed) instead of NT score tiles held in registers and fully unrolled loops: ~100 VGPRs, 8 waves per
// workgroup and two workgroups (78 KiB of K / V images each) per CU.
template <int NT, bool CAUSAL, int NW>
__global__ __launch_bounds__(64 * NW, 4) void attn_fwd_long2_kernel(const bf16_t* qkv, bf16_t* out, int S, int W, int H, float scale,
                                                                    float* lse /* [batch * H * S] or null: -(m log2 e + log2 l) per query row */) {
    constexpr int KS = (NT + 1) / 2;
    constexpr bool ODD = (NT & 1) != 0;
    constexpr int IMG = NT * 16 * LDR;
    extern __shared__ __attribute__((aligned(16))) bf16_t lds_f2[];
    bf16_t* Ks = lds_f2;
    bf16_t* Vs = Ks + IMG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int64_t ld = 3 * (int64_t)W;
    const bf16_t* qb = qkv + (int64_t)b * S * ld + h * HD;
    // the wave's first query tile: fragments requested in front of the staging; the next tile's under this tile's arithmetic (a query
    // tile used to start with an exposed round trip to global memory)
    bf16x8 q0 = row_frag_global(qb, ld, wave, 0, lane, S), q1 = row_frag_global(qb, ld, wave, 1, lane, S);
    stage_head_block(Ks, qb + W, ld, S, NT * 16, tid, 64 * NW);
    stage_head_block(Vs, qb + 2 * W, ld, S, NT * 16, tid, 64 * NW);
    __syncthreads();
    const int g = lane >> 4, c16 = lane & 15;
    const int n_t = (S + 15) >> 4;
    for (int it = wave; it < n_t; it += NW) {
        bf16x8 nq0 = q0, nq1 = q1;
        if (it + NW < n_t) { nq0 = row_frag_global(qb, ld, it + NW, 0, lane, S); nq1 = row_frag_global(qb, ld, it + NW, 1, lane, S); }
        const int i = it * 16 + c16;
        const int jt_end = CAUSAL ? min(n_t, it + 1) : n_t;
        // Unscaled scores: the maximum commutes with the (positive) scale, and P = 2^((s - max) c) takes one fma and one v_exp per pair.
        // Only the last key tile (keys beyond the sequence) and, under the causal mask, the diagonal tile hold masked pairs: the others
        // skip the compare / select.  (Measured VALU-bound: scale, two compares and a select per score in BOTH sweeps, 3 350 vector
        // instructions per wave against 104 MFMAs.)
        const float c = scale * 1.4426950408889634f;
        auto scores = [&](int jt) -> f32x4 {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = MFMA16(row_frag_lds(Ks, jt, 0, lane), q0, a);
            a = MFMA16(row_frag_lds(Ks, jt, 1, lane), q1, a);
            if (jt * 16 + 16 > S || (CAUSAL && jt == it)) {   // wave-uniform
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = jt * 16 + 4 * g + r;
                    a[r] = (j < S && (!CAUSAL || j <= i)) ? a[r] : -INFINITY;
                }
            }
            return a;
        };
        float m = -INFINITY;
        for (int jt = 0; jt < jt_end; ++jt) {
            const f32x4 a = scores(jt);
            m = fmaxf(fmaxf(m, fmaxf(a[0], a[1])), fmaxf(a[2], a[3]));
        }
        m = group_max(m);
        const float nmc = -m * c;
        float l = 0.f;
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int s = 0; s < KS; ++s) {
            if (2 * s >= jt_end) break;
            f32x4 pr[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int jt = 2 * s + u;
                pr[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (jt >= jt_end) continue;
                const f32x4 a = scores(jt);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pr[u][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(a[r], c, nmc));   // 2^(-inf) = 0 for masked keys
                    l += pr[u][r];
                }
            }
            const bf16x8 pf = pack_frag(pr[0], pr[1]);
            const bool hi_valid = !(ODD && s == KS - 1);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 vt = hi_valid ? tr_frag<true>(Vs, s, 16 * dt, lane) : tr_frag<false>(Vs, s, 16 * dt, lane);
                o[dt] = MFMA16(vt, pf, o[dt]);
            }
        }
        l = group_sum(l);
        const float inv = 1.0f / l;
        if (i < S) {
            bf16_t* op = out + ((int64_t)b * S + i) * W + h * HD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) io<bf16_t>::st4(op + 16 * dt, o[dt] * inv);
            if (lse && g == 0) lse[(int64_t)blockIdx.x * S + i] = __builtin_amdgcn_logf(inv) + nmc;   // log2(1 / l) - max c: P_ij = 2^(s_ij c + lse_i)
        }
        q0 = nq0; q1 = nq1;
    }
}

// ------------------------------------------------------------------------------------------------ backward, long sequences
// Same two passes as above, but pass 1 does not keep the scores and dP of all NT key tiles of a query tile in registers (140
// VGPRs at NT = 17, which holds the kernel above to one wave per SIMD and 6.1 ms per ViT-L/14 layer): it sweeps the key tiles
// three times - row max; row sum and delta = sum_j p_ij dp_ij; dQ - recomputing the two 16x16 MFMA products per key tile each
// time (MFMA work is negligible here).  V is staged in LDS as a fourth image (4 x 272 x 144 B + statistics = 156 KiB), so
// every fragment comes from LDS and the register budget allows NW = 16 waves per workgroup (4 per SIMD).
//
// STATS = true (round 3, the flash-attention form): the forward hands over lse_i = -(m_i log2 e + log2 l_i) and its output O, so
// P_ij = 2^(s_ij c + lse_i) needs no row maximum and no row sum, and delta_i = sum_j p_ij dp_ij = dO_i . O_i is a 64-term dot product per
// query row.  Pass 1 is then ONE sweep over the key tiles (scores, dP, dS, dQ) instead of three, pass 2 reads lse / delta from LDS; masks
// are applied on the tiles that contain a masked pair only, rows beyond the sequence carry lse = -inf (P = 0).  ViT-L/14 (S = 257, 16
// heads, local batch 512): 2.12 -> see profiles/r03_attention_l14_times.txt.
template <int NT, bool CAUSAL, int NW, bool STATS>
__global__ __launch_bounds__(64 * NW, NW / 4) void attn_bwd_long2_kernel(const bf16_t* qkv, const bf16_t* d_out, bf16_t* d_qkv, int S, int W, int H, float scale,
                                                                         float* cs_part /* [batch][3 W] or null */, const bf16_t* fwd_out, const float* lse) {
    constexpr int KS = (NT + 1) / 2;
    constexpr bool ODD = (NT & 1) != 0;
    constexpr int IMG = NT * 16 * LDR;
    extern __shared__ __attribute__((aligned(16))) bf16_t lds_b2[];
    bf16_t* Ks = lds_b2;
    bf16_t* Qs = Ks + IMG;
    bf16_t* Os = Qs + IMG;                  // dO
    bf16_t* Vs = Os + IMG;
    float* st_m = (float*)(Vs + IMG);
    float* st_il = st_m + NT * 16;
    float* st_dl = st_il + NT * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int64_t ld = 3 * (int64_t)W;
    const bf16_t* qb = qkv + (int64_t)b * S * ld + h * HD;
    const bf16_t* dob = d_out + (int64_t)b * S * W + h * HD;
    bf16_t* dqb = d_qkv + (int64_t)b * S * ld + h * HD;
    stage_head_block(Qs, qb, ld, S, NT * 16, tid, 64 * NW);
    stage_head_block(Ks, qb + W, ld, S, NT * 16, tid, 64 * NW);
    stage_head_block(Vs, qb + 2 * W, ld, S, NT * 16, tid, 64 * NW);
    stage_head_block(Os, dob, W, S, NT * 16, tid, 64 * NW);
    __syncthreads();
    const int g = lane >> 4, c16 = lane & 15;
    const int n_t = (S + 15) >> 4;

    f32x4 csq[4], csk[4], csv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) csq[dt] = csk[dt] = csv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float c = scale * 1.4426950408889634f;   // exp(x scale) = 2^(x c)
    // ---------------- pass 1: lane = query row i
    if constexpr (STATS) {
        for (int it = wave; it < n_t; it += NW) {
            const bf16x8 q0 = row_frag_lds(Qs, it, 0, lane), q1 = row_frag_lds(Qs, it, 1, lane);
            const bf16x8 g0 = row_frag_lds(Os, it, 0, lane), g1 = row_frag_lds(Os, it, 1, lane);
            const int i = it * 16 + c16;
            const bool live = i < S;
            const int jt_end = CAUSAL ? min(n_t, it + 1) : n_t;
            // delta_i = dO_i . O_i: this lane's 16 of the 64 head dimensions (the fragment's own), summed over the four lane groups
            const bf16x8 o0 = row_frag_global(fwd_out + (int64_t)b * S * W + h * HD, W, it, 0, lane, S),
                         o1 = row_frag_global(fwd_out + (int64_t)b * S * W + h * HD, W, it, 1, lane, S);
            float dot = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                dot += bf16_to_f32((bf16_t)g0[e]) * bf16_to_f32((bf16_t)o0[e]) + bf16_to_f32((bf16_t)g1[e]) * bf16_to_f32((bf16_t)o1[e]);
            const float delta = group_sum(dot);        // rows beyond the sequence: their dO rows are zero in the LDS image
            const float ei = live ? lse[(int64_t)blockIdx.x * S + i] : -INFINITY;
            const float nds = -delta * scale;
            if (g == 0) { st_m[i] = ei; st_dl[i] = live ? nds : 0.f; }
            f32x4 dq[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < KS; ++s) {
                if (2 * s >= jt_end) break;
                f32x4 ds[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int jt = 2 * s + u;
                    ds[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (jt >= jt_end) continue;
                    f32x4 a = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};
                    a = MFMA16(row_frag_lds(Ks, jt, 0, lane), q0, a);
                    a = MFMA16(row_frag_lds(Ks, jt, 1, lane), q1, a);
                    d = MFMA16(row_frag_lds(Vs, jt, 0, lane), g0, d);
                    d = MFMA16(row_frag_lds(Vs, jt, 1, lane), g1, d);
                    const bool edge = jt * 16 + 16 > S || (CAUSAL && jt == it);   // the only tiles with masked pairs (wave-uniform)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(a[r], c, ei));
                        if (edge) {
                            const int j = jt * 16 + 4 * g + r;
                            p = (j < S && (!CAUSAL || j <= i)) ? p : 0.f;
                        }
                        ds[u][r] = p * __builtin_fmaf(d[r], scale, nds);
                    }
                }
                const bf16x8 dsf = pack_frag(ds[0], ds[1]);
                const bool hi_valid = !(ODD && s == KS - 1);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8 kt = hi_valid ? tr_frag<true>(Ks, s, 16 * dt, lane) : tr_frag<false>(Ks, s, 16 * dt, lane);
                    dq[dt] = MFMA16(kt, dsf, dq[dt]);
                }
            }
            if (live) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) io<bf16_t>::st4(dqb + (int64_t)i * ld + 16 * dt + 4 * g, dq[dt]);
            }
            if (cs_part) cs_add(csq, dq, live);
        }
    } else
    for (int it = wave; it < n_t; it += NW) {
        const bf16x8 q0 = row_frag_lds(Qs, it, 0, lane), q1 = row_frag_lds(Qs, it, 1, lane);
        const bf16x8 g0 = row_frag_lds(Os, it, 0, lane), g1 = row_frag_lds(Os, it, 1, lane);
        const int i = it * 16 + c16;
        const int jt_end = CAUSAL ? min(n_t, it + 1) : n_t;   // key tiles past the diagonal are masked out entirely
        auto scores = [&](int jt) -> f32x4 {                  // scaled, -inf where masked
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = MFMA16(row_frag_lds(Ks, jt, 0, lane), q0, a);
            a = MFMA16(row_frag_lds(Ks, jt, 1, lane), q1, a);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = jt * 16 + 4 * g + r;
                const bool ok = j < S && (!CAUSAL || j <= i);
                a[r] = ok ? a[r] * scale : -INFINITY;
            }
            return a;
        };
        auto dprobs = [&](int jt) -> f32x4 {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
            d = MFMA16(row_frag_lds(Vs, jt, 0, lane), g0, d);
            d = MFMA16(row_frag_lds(Vs, jt, 1, lane), g1, d);
            return d;
        };
        float m = -INFINITY;
        for (int jt = 0; jt < jt_end; ++jt) {
            const f32x4 a = scores(jt);
            m = fmaxf(fmaxf(m, fmaxf(a[0], a[1])), fmaxf(a[2], a[3]));
        }
        m = group_max(m);
        float l = 0.f, num = 0.f;
        for (int jt = 0; jt < jt_end; ++jt) {
            const f32x4 a = scores(jt), d = dprobs(jt);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(a[r] - m);   // exp(-inf) = 0 for masked keys
                l += p;
                num += p * d[r];
            }
        }
        l = group_sum(l);
        const float inv = 1.0f / l;
        const float delta = group_sum(num) * inv;
        if (g == 0) {
            const bool live = i < S;
            st_m[i] = live ? m : 0.f;
            st_il[i] = live ? inv : 0.f;
            st_dl[i] = live ? delta : 0.f;
        }
        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < KS; ++s) {
            if (2 * s >= jt_end) break;
            f32x4 ds[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int jt = 2 * s + u;
                ds[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (jt >= jt_end) continue;
                const f32x4 a = scores(jt), d = dprobs(jt);
#pragma unroll
                for (int r = 0; r < 4; ++r) ds[u][r] = __expf(a[r] - m) * inv * (d[r] - delta) * scale;
            }
            const bf16x8 dsf = pack_frag(ds[0], ds[1]);
            const bool hi_valid = !(ODD && s == KS - 1);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 kt = hi_valid ? tr_frag<true>(Ks, s, 16 * dt, lane) : tr_frag<false>(Ks, s, 16 * dt, lane);
                dq[dt] = MFMA16(kt, dsf, dq[dt]);
            }
        }
        if (i < S) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) io<bf16_t>::st4(dqb + (int64_t)i * ld + 16 * dt + 4 * g, dq[dt]);
        }
        if (cs_part) cs_add(csq, dq, i < S);
    }
    // tiles of padding queries (none when n_t == NT) must read as "no contribution" in pass 2
    for (int i = n_t * 16 + tid; i < NT * 16; i += 64 * NW) st_m[i] = st_il[i] = st_dl[i] = 0.f;
    __syncthreads();   // every query tile's statistics are in LDS

    // ---------------- pass 2: lane = key row j
    for (int jt = wave; jt < n_t; jt += NW) {
        const bf16x8 k0 = row_frag_lds(Ks, jt, 0, lane), k1 = row_frag_lds(Ks, jt, 1, lane);
        const bf16x8 v0 = row_frag_lds(Vs, jt, 0, lane), v1 = row_frag_lds(Vs, jt, 1, lane);
        const int j = jt * 16 + c16;
        f32x4 dv[4], dk[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dv[dt] = dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int s = 0; s < KS; ++s) {   // a run-time loop: unrolled over all 9 k-steps the compiler keeps hundreds of values live
            f32x4 pt[2], dst[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                pt[u] = dst[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int it = 2 * s + u;
                if (ODD && it >= NT) continue;
                if (it >= n_t) continue;            // query tile beyond the sequence: pass 1 wrote no statistics for it
                if (CAUSAL && it < jt) continue;    // query tile entirely in the past of this key tile: P^T = dS^T = 0
                f32x4 a = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};
                a = MFMA16(row_frag_lds(Qs, it, 0, lane), k0, a);
                a = MFMA16(row_frag_lds(Qs, it, 1, lane), k1, a);
                d = MFMA16(row_frag_lds(Os, it, 0, lane), v0, d);
                d = MFMA16(row_frag_lds(Os, it, 1, lane), v1, d);
                const f32x4 mm = *(const f32x4*)(st_m + it * 16 + 4 * g), dl = *(const f32x4*)(st_dl + it * 16 + 4 * g);
                if constexpr (STATS) {   // mm = lse (-inf for rows beyond the sequence: p = 0, their dO rows are zero), dl = -scale delta;
                                         // keys beyond the sequence only produce lanes that are never stored: mask the diagonal tile only
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(a[r], c, mm[r]));
                        if (CAUSAL && it == jt) p = (j <= it * 16 + 4 * g + r) ? p : 0.f;
                        pt[u][r] = p;
                        dst[u][r] = p * __builtin_fmaf(d[r], scale, dl[r]);
                    }
                } else {
                    const f32x4 il = *(const f32x4*)(st_il + it * 16 + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = it * 16 + 4 * g + r;
                        const bool ok = i < S && j < S && (!CAUSAL || j <= i);
                        const float p = ok ? __expf(a[r] * scale - mm[r]) * il[r] : 0.f;
                        pt[u][r] = p;
                        dst[u][r] = ok ? p * (d[r] - dl[r]) * scale : 0.f;   // not 0 * (d - dl): d, dl of a masked pair need not be finite
                    }
                }
            }
            if (CAUSAL && 2 * s + 1 < jt) continue;
            if (2 * s >= n_t) continue;
            const bf16x8 pf = pack_frag(pt[0], pt[1]), dsf = pack_frag(dst[0], dst[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bool hi_valid = !(ODD && s == KS - 1);
                const bf16x8 ot = hi_valid ? tr_frag<true>(Os, s, 16 * dt, lane) : tr_frag<false>(Os, s, 16 * dt, lane);
                const bf16x8 qt = hi_valid ? tr_frag<true>(Qs, s, 16 * dt, lane) : tr_frag<false>(Qs, s, 16 * dt, lane);
                dv[dt] = MFMA16(ot, pf, dv[dt]);
                dk[dt] = MFMA16(qt, dsf, dk[dt]);
            }
        }
        if (j < S) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                io<bf16_t>::st4(dqb + (int64_t)j * ld + W + 16 * dt + 4 * g, dk[dt]);
                io<bf16_t>::st4(dqb + (int64_t)j * ld + 2 * W + 16 * dt + 4 * g, dv[dt]);
            }
        }
        if (cs_part) { cs_add(csk, dk, j < S); cs_add(csv, dv, j < S); }
    }
    if (cs_part) {   // per-wave sums over its tiles -> LDS -> the head's 192 columns for image b, waves added in a fixed order
        cs_rows(csq); cs_rows(csk); cs_rows(csv);
        __syncthreads();                       // the LDS images are dead
        float* red = (float*)lds_b2;           // [NW][192]
        if (c16 == 0) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                *(f32x4*)(red + wave * 192 + 16 * dt + 4 * g) = csq[dt];
                *(f32x4*)(red + wave * 192 + 64 + 16 * dt + 4 * g) = csk[dt];
                *(f32x4*)(red + wave * 192 + 128 + 16 * dt + 4 * g) = csv[dt];
            }
        }
        __syncthreads();
        if (tid < 192) {
            float v = red[tid];
#pragma unroll
            for (int w2 = 1; w2 < NW; ++w2) v += red[w2 * 192 + tid];
            cs_part[(int64_t)b * 3 * W + (tid >> 6) * W + h * HD + (tid & 63)] = v;
        }
    }
}

// column sums of one tile (values as stored: bf16-rounded; rows beyond the sequence excluded) added to the wave's 64 floats in LDS: the
// running sums of the kernels above cost 16 registers per output, which the kernel below does not have
__device__ __forceinline__ void cs_tile_to_lds(float* dst, const f32x4 (&v)[4], bool live, int c16, int g) {
    f32x4 t[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[dt][r] = live ? bf16_to_f32(f32_to_bf16(v[dt][r])) : 0.f;
    cs_rows(t);
    if (c16 == 0) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *(f32x4*)(dst + 16 * dt + 4 * g) += t[dt];
    }
}

// ------------------------------------------------------------------------------------------------ backward, long sequences, two LDS phases
// The statistics form of the kernel above with HALF its LDS: pass 1 (lane = query row) only needs the K and V images, pass 2 (lane = key
// row) only the Q and dO images, and a wave's own row fragments (Q / dO in pass 1, K / V in pass 2) come straight from global memory, one
// tile ahead.  76 KiB instead of 139 KiB and <= 128 registers: TWO workgroups (16 waves, 4 per SIMD) per CU, so that one head's staging,
// barriers and the uneven deal of 17 tiles to 8 waves are covered by the other head's waves.  lse and delta = dO . O of every row are taken
// once per head by all threads (8 lanes per row, coalesced) into LDS instead of per query tile out of fragment loads.
template <int NT, bool CAUSAL>
__global__ __launch_bounds__(512, 4) void attn_bwd_long3_kernel(const bf16_t* qkv, const bf16_t* d_out, bf16_t* d_qkv, int S, int W, int H, float scale,
                                                                float* cs_part /* [batch][3 W] or null */, const bf16_t* fwd_out, const float* lse) {
    constexpr int NW = 8, CW = 4;           // waves; waves that share a lone last tile
    constexpr int KS = (NT + 1) / 2;
    constexpr bool ODD = (NT & 1) != 0;
    constexpr int IMG = NT * 16 * LDR;
    extern __shared__ __attribute__((aligned(16))) bf16_t lds_b3[];
    bf16_t* Ia = lds_b3;                    // pass 1: K, pass 2: Q
    bf16_t* Ib = Ia + IMG;                  // pass 1: V, pass 2: dO
    float* st_e = (float*)(Ib + IMG);       // lse_i (-inf beyond the sequence)
    float* st_dl = st_e + NT * 16;          // -scale delta_i
    float* red = st_dl + NT * 16;           // [NW][192] column sums of dq | dk | dv
    float* part = red + NW * 192;           // [CW][128] partial rows of the lone tile (dq; then dk | dv)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int64_t ld = 3 * (int64_t)W;
    const bf16_t* qb = qkv + (int64_t)b * S * ld + h * HD;
    const bf16_t* dob = d_out + (int64_t)b * S * W + h * HD;
    const bf16_t* fob = fwd_out + (int64_t)b * S * W + h * HD;
    bf16_t* dqb = d_qkv + (int64_t)b * S * ld + h * HD;
    const int g = lane >> 4, c16 = lane & 15;
    const int n_t = (S + 15) >> 4;
    const float c = scale * 1.4426950408889634f;   // exp(x scale) = 2^(x c)
    // A last tile of ONE row that would be some wave's extra tile (S = 257: 16 whole tiles + the class token's row; dealt out whole it
    // makes wave 0 work three tiles where the others work two): waves 0 .. CW-1 take every CW-th pair of its inner tiles after their own
    // tiles, the partial rows meet in LDS behind the pass's barrier.
    const bool lone = n_t > NW && n_t % NW == 1 && S % 16 == 1;
    const int n_own = lone ? n_t - 1 : n_t;

    // the wave's first query tile: fragments requested in front of the staging
    bf16x8 q0 = row_frag_global(qb, ld, wave, 0, lane, S), q1 = row_frag_global(qb, ld, wave, 1, lane, S);
    bf16x8 g0 = row_frag_global(dob, W, wave, 0, lane, S), g1 = row_frag_global(dob, W, wave, 1, lane, S);
    stage_head_block(Ia, qb + W, ld, S, NT * 16, tid, 64 * NW);
    stage_head_block(Ib, qb + 2 * W, ld, S, NT * 16, tid, 64 * NW);
    for (int x = tid; x < NW * 192; x += 64 * NW) red[x] = 0.f;
    for (int r = tid >> 3; r < NT * 16; r += 8 * NW) {   // statistics: 8 lanes per row, 16 bytes of dO and O each
        float dot = 0.f;
        if (r < S) {
            const uint4 a = *(const uint4*)(dob + (int64_t)r * W + (tid & 7) * 8), o = *(const uint4*)(fob + (int64_t)r * W + (tid & 7) * 8);
            const unsigned av[4] = {a.x, a.y, a.z, a.w}, ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                dot += __uint_as_float(av[e] << 16) * __uint_as_float(ov[e] << 16) + __uint_as_float(av[e] & 0xffff0000u) * __uint_as_float(ov[e] & 0xffff0000u);
        }
        dot += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(dot), 0xB1, 0xF, 0xF, true));    // lanes xor 1
        dot += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(dot), 0x4E, 0xF, 0xF, true));    // lanes xor 2
        dot += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(dot), 0x141, 0xF, 0xF, true));   // the other quad of the 8 lanes
        if ((tid & 7) == 0) {
            st_e[r] = r < S ? lse[(int64_t)blockIdx.x * S + r] : -INFINITY;
            st_dl[r] = r < S ? -dot * scale : 0.f;
        }
    }
    __syncthreads();

    // ---------------- pass 1: lane = query row i; Ia = K, Ib = V
    // dq of query tile `it` over the key-tile pairs s0, s0 + sstep, ...
    auto pass1 = [&](int it, const bf16x8& q0, const bf16x8& q1, const bf16x8& g0, const bf16x8& g1, int s0, int sstep, f32x4 (&dq)[4]) {
        const int i = it * 16 + c16;
        const int jt_end = CAUSAL ? min(n_t, it + 1) : n_t;
        const float ei = st_e[i], nds = st_dl[i];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int s = s0; s < KS; s += sstep) {
            if (2 * s >= jt_end) break;
            f32x4 ds[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int jt = 2 * s + u;
                ds[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (jt >= jt_end) continue;
                f32x4 a = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};
                a = MFMA16(row_frag_lds(Ia, jt, 0, lane), q0, a);
                a = MFMA16(row_frag_lds(Ia, jt, 1, lane), q1, a);
                d = MFMA16(row_frag_lds(Ib, jt, 0, lane), g0, d);
                d = MFMA16(row_frag_lds(Ib, jt, 1, lane), g1, d);
                const bool edge = jt * 16 + 16 > S || (CAUSAL && jt == it);   // the only tiles with masked pairs (wave-uniform)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p = __builtin_amdgcn_exp2f(__builtin_fmaf(a[r], c, ei));
                    if (edge) {
                        const int j = jt * 16 + 4 * g + r;
                        p = (j < S && (!CAUSAL || j <= i)) ? p : 0.f;
                    }
                    ds[u][r] = p * __builtin_fmaf(d[r], scale, nds);
                }
            }
            const bf16x8 dsf = pack_frag(ds[0], ds[1]);
            const bool hi_valid = !(ODD && s == KS - 1);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 kt = hi_valid ? tr_frag<true>(Ia, s, 16 * dt, lane) : tr_frag<false>(Ia, s, 16 * dt, lane);
                dq[dt] = MFMA16(kt, dsf, dq[dt]);
            }
        }
    };
    for (int it = wave; it < n_own; it += NW) {
        bf16x8 nq0 = q0, nq1 = q1, ng0 = g0, ng1 = g1;
        const int itn = it + NW < n_own ? it + NW : (lone && wave < CW ? n_t - 1 : -1);   // the next own tile, then the lone tile
        if (itn >= 0) {   // its fragments, under this tile's arithmetic
            nq0 = row_frag_global(qb, ld, itn, 0, lane, S); nq1 = row_frag_global(qb, ld, itn, 1, lane, S);
            ng0 = row_frag_global(dob, W, itn, 0, lane, S); ng1 = row_frag_global(dob, W, itn, 1, lane, S);
        }
        f32x4 dq[4];
        pass1(it, q0, q1, g0, g1, 0, 1, dq);
        const int i = it * 16 + c16;
        if (i < S) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) io<bf16_t>::st4(dqb + (int64_t)i * ld + 16 * dt + 4 * g, dq[dt]);
        }
        if (cs_part) cs_tile_to_lds(red + wave * 192, dq, i < S, c16, g);
        q0 = nq0; q1 = nq1; g0 = ng0; g1 = ng1;
    }
    if (lone && wave < CW) {   // this wave's share of the lone query row (lane c16 == 0 of every lane group holds its 16 of the 64 columns)
        f32x4 dq[4];
        pass1(n_t - 1, q0, q1, g0, g1, wave, CW, dq);
        if (c16 == 0) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *(f32x4*)(part + wave * 128 + 16 * dt + 4 * g) = dq[dt];
        }
    }
    // the wave's first key tile: K / V rows from global, requested in front of the second staging
    bf16x8 k0 = row_frag_global(qb + W, ld, wave, 0, lane, S), k1 = row_frag_global(qb + W, ld, wave, 1, lane, S);
    bf16x8 v0 = row_frag_global(qb + 2 * W, ld, wave, 0, lane, S), v1 = row_frag_global(qb + 2 * W, ld, wave, 1, lane, S);
    __syncthreads();   // every wave is done with the K and V images
    stage_head_block(Ia, qb, ld, S, NT * 16, tid, 64 * NW);
    stage_head_block(Ib, dob, W, S, NT * 16, tid, 64 * NW);
    if (lone && tid < 64) {   // the lone query row: partial rows in wave order; wave 0 owns red[0 .. 191]
        float v = part[tid];
#pragma unroll
        for (int w2 = 1; w2 < CW; ++w2) v += part[w2 * 128 + tid];
        const bf16_t o = f32_to_bf16(v);
        dqb[(int64_t)(S - 1) * ld + tid] = o;
        if (cs_part) red[tid] += bf16_to_f32(o);
    }
    __syncthreads();

    // ---------------- pass 2: lane = key row j; Ia = Q, Ib = dO
    auto pass2 = [&](int jt, const bf16x8& k0, const bf16x8& k1, const bf16x8& v0, const bf16x8& v1, int s0, int sstep, f32x4 (&dk)[4], f32x4 (&dv)[4]) {
        const int j = jt * 16 + c16;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dv[dt] = dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int s = s0; s < KS; s += sstep) {
            f32x4 pt[2], dst[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                pt[u] = dst[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int it = 2 * s + u;
                if (ODD && it >= NT) continue;
                if (it >= n_t) continue;
                if (CAUSAL && it < jt) continue;    // query tile entirely in the past of this key tile: P^T = dS^T = 0
                f32x4 a = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};
                a = MFMA16(row_frag_lds(Ia, it, 0, lane), k0, a);
                a = MFMA16(row_frag_lds(Ia, it, 1, lane), k1, a);
                d = MFMA16(row_frag_lds(Ib, it, 0, lane), v0, d);
                d = MFMA16(row_frag_lds(Ib, it, 1, lane), v1, d);
                const f32x4 mm = *(const f32x4*)(st_e + it * 16 + 4 * g), dl = *(const f32x4*)(st_dl + it * 16 + 4 * g);
                // rows beyond the sequence carry lse = -inf (p = 0); keys beyond it only produce lanes that are never stored
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p = __builtin_amdgcn_exp2f(__builtin_fmaf(a[r], c, mm[r]));
                    if (CAUSAL && it == jt) p = (j <= it * 16 + 4 * g + r) ? p : 0.f;
                    pt[u][r] = p;
                    dst[u][r] = p * __builtin_fmaf(d[r], scale, dl[r]);
                }
            }
            if (CAUSAL && 2 * s + 1 < jt) continue;
            if (2 * s >= n_t) continue;
            const bf16x8 pf = pack_frag(pt[0], pt[1]), dsf = pack_frag(dst[0], dst[1]);
            const bool hi_valid = !(ODD && s == KS - 1);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 ot = hi_valid ? tr_frag<true>(Ib, s, 16 * dt, lane) : tr_frag<false>(Ib, s, 16 * dt, lane);
                const bf16x8 qt = hi_valid ? tr_frag<true>(Ia, s, 16 * dt, lane) : tr_frag<false>(Ia, s, 16 * dt, lane);
                dv[dt] = MFMA16(ot, pf, dv[dt]);
                dk[dt] = MFMA16(qt, dsf, dk[dt]);
            }
        }
    };
    for (int jt = wave; jt < n_own; jt += NW) {
        if (jt != wave) {   // (no room to hold the next tile's fragments beside this tile's: the CU's other waves cover the round trip)
            k0 = row_frag_global(qb + W, ld, jt, 0, lane, S); k1 = row_frag_global(qb + W, ld, jt, 1, lane, S);
            v0 = row_frag_global(qb + 2 * W, ld, jt, 0, lane, S); v1 = row_frag_global(qb + 2 * W, ld, jt, 1, lane, S);
        }
        f32x4 dv[4], dk[4];
        pass2(jt, k0, k1, v0, v1, 0, 1, dk, dv);
        const int j = jt * 16 + c16;
        if (j < S) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                io<bf16_t>::st4(dqb + (int64_t)j * ld + W + 16 * dt + 4 * g, dk[dt]);
                io<bf16_t>::st4(dqb + (int64_t)j * ld + 2 * W + 16 * dt + 4 * g, dv[dt]);
            }
        }
        if (cs_part) { cs_tile_to_lds(red + wave * 192 + 64, dk, j < S, c16, g); cs_tile_to_lds(red + wave * 192 + 128, dv, j < S, c16, g); }
    }
    if (lone && wave < CW) {   // this wave's share of the lone key row
        k0 = row_frag_global(qb + W, ld, n_t - 1, 0, lane, S); k1 = row_frag_global(qb + W, ld, n_t - 1, 1, lane, S);
        v0 = row_frag_global(qb + 2 * W, ld, n_t - 1, 0, lane, S); v1 = row_frag_global(qb + 2 * W, ld, n_t - 1, 1, lane, S);
        f32x4 dv[4], dk[4];
        pass2(n_t - 1, k0, k1, v0, v1, wave, CW, dk, dv);
        if (c16 == 0) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                *(f32x4*)(part + wave * 128 + 16 * dt + 4 * g) = dk[dt];
                *(f32x4*)(part + wave * 128 + 64 + 16 * dt + 4 * g) = dv[dt];
            }
        }
    }
    if (lone || cs_part) __syncthreads();
    if (lone && tid < 128) {   // the lone key row: dk (threads 0 .. 63, wave 0) and dv (64 .. 127, wave 1), partial rows in wave order
        float v = part[tid];
#pragma unroll
        for (int w2 = 1; w2 < CW; ++w2) v += part[w2 * 128 + tid];
        const bf16_t o = f32_to_bf16(v);
        dqb[(int64_t)(S - 1) * ld + W + (tid >> 6) * W + (tid & 63)] = o;
        if (cs_part) red[(tid >> 6) * 192 + 64 + tid] += bf16_to_f32(o);   // each into its own wave's row: columns 64 + tid (dk), 128 + tid - 64 (dv)
    }
    if (cs_part) {   // the waves' sums over their tiles -> the head's 192 columns for image b, waves added in a fixed order
        if (lone) __syncthreads();
        if (tid < 192) {
            float v = red[tid];
#pragma unroll
            for (int w2 = 1; w2 < NW; ++w2) v += red[w2 * 192 + tid];
            cs_part[(int64_t)b * 3 * W + (tid >> 6) * W + h * HD + (tid & 63)] = v;
        }
    }
}

template <typename K>
int reserve_lds(K kernel, size_t bytes) {
    if (bytes > 65536) {
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return sc_set_error((int)e, "attention(long): cannot reserve %zu bytes of LDS: %s", bytes, hipGetErrorString(e));
    }
    return SC_OK;
}

constexpr int NT_LONG = 17;   // 272 rows: S <= 272

template <int NT, int NW>
int launch_fwd_block(const void* qkv, void* out, int64_t batch, int64_t seq, int64_t width, int64_t heads, int causal, hipStream_t st) {
    const size_t lds = (size_t)2 * NT * 16 * LDR * sizeof(bf16_t);
    const dim3 grid((unsigned)(batch * heads));
    if (causal) {
        SC_TRY(reserve_lds(attn_fwd_long_kernel<NT, true, NW>, lds));
        hipLaunchKernelGGL((attn_fwd_long_kernel<NT, true, NW>), grid, dim3(64 * NW), lds, st, (const bf16_t*)qkv, (bf16_t*)out, (int)seq, (int)width, (int)heads, 0.125f);
    } else {
        SC_TRY(reserve_lds(attn_fwd_long_kernel<NT, false, NW>, lds));
        hipLaunchKernelGGL((attn_fwd_long_kernel<NT, false, NW>), grid, dim3(64 * NW), lds, st, (const bf16_t*)qkv, (bf16_t*)out, (int)seq, (int)width, (int)heads, 0.125f);
    }
    SC_CHECK_LAUNCH();
    return SC_OK;
}
template <int NT, int NW>
int launch_bwd_block(const void* qkv, const void* d_out, void* d_qkv, int64_t batch, int64_t seq, int64_t width, int64_t heads, int causal, float* cs_part,
                     hipStream_t st) {
    const size_t lds = ((size_t)3 * NT * 16 * LDR) * sizeof(bf16_t) + (size_t)3 * NT * 16 * sizeof(float);
    const dim3 grid((unsigned)(batch * heads));
    if (causal) {
        SC_TRY(reserve_lds(attn_bwd_long_kernel<NT, true, NW>, lds));
        hipLaunchKernelGGL((attn_bwd_long_kernel<NT, true, NW>), grid, dim3(64 * NW), lds, st, (const bf16_t*)qkv, (const bf16_t*)d_out, (bf16_t*)d_qkv, (int)seq,
                           (int)width, (int)heads, 0.125f, cs_part);
    } else {
        SC_TRY(reserve_lds(attn_bwd_long_kernel<NT, false, NW>, lds));
        hipLaunchKernelGGL((attn_bwd_long_kernel<NT, false, NW>), grid, dim3(64 * NW), lds, st, (const bf16_t*)qkv, (const bf16_t*)d_out, (bf16_t*)d_qkv, (int)seq,
                           (int)width, (int)heads, 0.125f, cs_part);
    }
    SC_CHECK_LAUNCH();
    return SC_OK;
}

template <int NT, int NW>
int launch_fwd_long2(const void* qkv, void* out, int64_t batch, int64_t seq, int64_t width, int64_t heads, int causal, hipStream_t st, float* lse = nullptr) {
    const size_t lds = (size_t)2 * NT * 16 * LDR * sizeof(bf16_t);
    const dim3 grid((unsigned)(batch * heads));
    if (causal) {
        SC_TRY(reserve_lds(attn_fwd_long2_kernel<NT, true, NW>, lds));
        hipLaunchKernelGGL((attn_fwd_long2_kernel<NT, true, NW>), grid, dim3(64 * NW), lds, st, (const bf16_t*)qkv, (bf16_t*)out, (int)seq, (int)width, (int)heads, 0.125f, lse);
    } else {
        SC_TRY(reserve_lds(attn_fwd_long2_kernel<NT, false, NW>, lds));
        hipLaunchKernelGGL((attn_fwd_long2_kernel<NT, false, NW>), grid, dim3(64 * NW), lds, st, (const bf16_t*)qkv, (bf16_t*)out, (int)seq, (int)width, (int)heads, 0.125f, lse);
    }
    SC_CHECK_LAUNCH();
    return SC_OK;
}

template <int NT, int NW>
int launch_bwd_long2(const void* qkv, const void* d_out, void* d_qkv, int64_t batch, int64_t seq, int64_t width, int64_t heads, int causal, float* cs_part,
                     hipStream_t st, const void* fwd_out = nullptr, const float* lse = nullptr) {
    const size_t lds = ((size_t)4 * NT * 16 * LDR) * sizeof(bf16_t) + (size_t)3 * NT * 16 * sizeof(float);
    const dim3 grid((unsigned)(batch * heads));
#define BWD_L2(C, ST)                                                                                                                                   \
    do {                                                                                                                                                \
        SC_TRY(reserve_lds(attn_bwd_long2_kernel<NT, C, NW, ST>, lds));                                                                                 \
        hipLaunchKernelGGL((attn_bwd_long2_kernel<NT, C, NW, ST>), grid, dim3(64 * NW), lds, st, (const bf16_t*)qkv, (const bf16_t*)d_out, (bf16_t*)d_qkv, \
                           (int)seq, (int)width, (int)heads, 0.125f, cs_part, (const bf16_t*)fwd_out, lse);                                             \
    } while (0)
    const bool stats = fwd_out != nullptr && lse != nullptr;
    if (causal) { if (stats) BWD_L2(true, true); else BWD_L2(true, false); }
    else { if (stats) BWD_L2(false, true); else BWD_L2(false, false); }
#undef BWD_L2
    SC_CHECK_LAUNCH();
    return SC_OK;
}

template <int NT>
int launch_bwd_long3(const void* qkv, const void* d_out, void* d_qkv, int64_t batch, int64_t seq, int64_t width, int64_t heads, int causal, float* cs_part,
                     hipStream_t st, const void* fwd_out, const float* lse) {
    const size_t lds = ((size_t)2 * NT * 16 * LDR) * sizeof(bf16_t) + (size_t)2 * NT * 16 * sizeof(float) + (size_t)8 * 192 * sizeof(float) + (size_t)4 * 128 * sizeof(float);
    const dim3 grid((unsigned)(batch * heads));
    if (causal) {
        SC_TRY(reserve_lds(attn_bwd_long3_kernel<NT, true>, lds));
        hipLaunchKernelGGL((attn_bwd_long3_kernel<NT, true>), grid, dim3(512), lds, st, (const bf16_t*)qkv, (const bf16_t*)d_out, (bf16_t*)d_qkv, (int)seq, (int)width,
                           (int)heads, 0.125f, cs_part, (const bf16_t*)fwd_out, lse);
    } else {
        SC_TRY(reserve_lds(attn_bwd_long3_kernel<NT, false>, lds));
        hipLaunchKernelGGL((attn_bwd_long3_kernel<NT, false>), grid, dim3(512), lds, st, (const bf16_t*)qkv, (const bf16_t*)d_out, (bf16_t*)d_qkv, (int)seq, (int)width,
                           (int)heads, 0.125f, cs_part, (const bf16_t*)fwd_out, lse);
    }
    SC_CHECK_LAUNCH();
    return SC_OK;
}

}  // namespace

// bf16, one workgroup per head, one wave per 16-row tile of the head where that fits (seq <= 64: 4 waves, <= 80: 5 waves) and
// 4 waves striding over the tiles up to seq = 272.  The three (forward: two) LDS images are shared by the waves of a
// workgroup: ~28 KiB per head at seq <= 64, so five workgroups = 20 waves fit a CU, where the one-wave-per-head kernels of
// attention_mfma.hip (private images, 110 KiB per 4-wave workgroup) are held to ONE wave per SIMD by LDS capacity.
// Returns 1 when the shape is not covered.
// lse (fp32 [batch * heads * seq]) is written by the kernel the long sequences take (80 < seq <= 272) and left alone otherwise:
// sc_attention_long_uses_stats says which, so that the backward knows whether the statistics exist
bool sc_attention_long_uses_stats(int64_t seq) {
    static const bool old_long = [] { const char* e = sc_debug_env("SC_ATTENTION_LONG_BWD"); return e && e[0] == '1'; }();
    return seq > 80 && seq <= NT_LONG * 16 && !old_long;
}

int sc_attention_long_fwd(const void* qkv, void* out, int64_t batch, int64_t seq, int64_t width, int64_t heads, int causal, hipStream_t st, float* lse) {
    if (seq > NT_LONG * 16) return 1;
    static const bool short2 = [] { const char* e = sc_debug_env("SC_ATTENTION_SHORT"); return e && e[0] == '2'; }();
    if (short2 && seq <= 64) return launch_fwd_long2<4, 4>(qkv, out, batch, seq, width, heads, causal, st);
    if (short2 && seq <= 80) return launch_fwd_long2<5, 5>(qkv, out, batch, seq, width, heads, causal, st);
    if (seq <= 64) return launch_fwd_block<4, 4>(qkv, out, batch, seq, width, heads, causal, st);
    if (seq <= 80) return launch_fwd_block<5, 5>(qkv, out, batch, seq, width, heads, causal, st);
    static const bool old_long = [] { const char* e = sc_debug_env("SC_ATTENTION_LONG_BWD"); return e && e[0] == '1'; }();   // =1: the register-resident variants (A/B)
    if (old_long) return launch_fwd_block<NT_LONG, 4>(qkv, out, batch, seq, width, heads, causal, st);
    return launch_fwd_long2<NT_LONG, 8>(qkv, out, batch, seq, width, heads, causal, st, lse);
}

int sc_attention_long_bwd(const void* qkv, const void* d_out, void* d_qkv, int64_t batch, int64_t seq, int64_t width, int64_t heads, int causal,
                          float* cs_part, hipStream_t st, const void* fwd_out, const float* lse) {
    if (seq > NT_LONG * 16) return 1;
    static const bool short2 = [] { const char* e = sc_debug_env("SC_ATTENTION_SHORT"); return e && e[0] == '2'; }();   // =2: recompute kernels for short sequences too (A/B)
    if (short2 && seq <= 64) return launch_bwd_long2<4, 4>(qkv, d_out, d_qkv, batch, seq, width, heads, causal, cs_part, st);
    if (short2 && seq <= 80) return launch_bwd_long2<5, 4>(qkv, d_out, d_qkv, batch, seq, width, heads, causal, cs_part, st);
    if (seq <= 64) return launch_bwd_block<4, 4>(qkv, d_out, d_qkv, batch, seq, width, heads, causal, cs_part, st);
    if (seq <= 80) return launch_bwd_block<5, 5>(qkv, d_out, d_qkv, batch, seq, width, heads, causal, cs_part, st);
    // 4 waves (one per SIMD, 512 registers each): the backward keeps the scores and dP of all 17 key tiles of a query tile in
    // registers; with 8 waves (256 registers) it spills ~400 VGPRs.  At ViT-L/14 scale this kernel is 35 % of the step
    // (6.1 ms per layer at local batch 512): a recompute-per-key-tile formulation is the next step for that model.
    static const bool old_long = [] { const char* e = sc_debug_env("SC_ATTENTION_LONG_BWD"); return e && e[0] == '1'; }();   // =1: the register-resident variant (A/B)
    if (old_long) return launch_bwd_block<NT_LONG, 4>(qkv, d_out, d_qkv, batch, seq, width, heads, causal, cs_part, st);
    const bool stats = fwd_out && lse && sc_attention_long_uses_stats(seq);
    static const bool one_phase = [] { const char* e = sc_debug_env("SC_ATTENTION_LONG3"); return e && e[0] == '0'; }();   // =0: all four images in LDS, one workgroup per CU (A/B)
    if (stats && !one_phase) return launch_bwd_long3<NT_LONG>(qkv, d_out, d_qkv, batch, seq, width, heads, causal, cs_part, st, fwd_out, lse);
    return launch_bwd_long2<NT_LONG, 8>(qkv, d_out, d_qkv, batch, seq, width, heads, causal, cs_part, st, stats ? fwd_out : nullptr, stats ? lse : nullptr);
}
