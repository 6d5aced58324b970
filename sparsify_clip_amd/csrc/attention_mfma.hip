// bf16 MFMA attention for the short sequences of the CLIP towers (S = 50 image tokens, S = 77 text tokens, head dim 64).
// One WAVE owns one (batch, head); 4 heads per workgroup, no workgroup barriers.  Scores, probabilities and their
// gradients never leave registers:
//   * S'[j][i] = K Q^T is computed with the operands swapped, so a lane holds ONE query row i and the keys j sit in the
//     accumulator registers: the softmax row reduction is in-register plus two cross-group shuffles;
//   * the accumulator tile is fed straight back as the B operand of the next MFMA (P V, dS K): that product sums over
//     the tile's ROW index, so no lane movement is needed - only the k order inside a 32-step is permuted
//     (key j = 32s + 4g + e for e < 4, 32s + 16 + 4g + (e-4) for e >= 4), and the other operand is fetched with the same
//     permutation by two ds_read_b64_tr_b16 of the row-major LDS image (cdna_hip_programming.md section 3 / T10);
//   * products that sum over the QUERY index (dV = P^T dO, dK = dS^T Q) use the non-swapped tile S[i][j] (lane = key),
//     recomputed in a second pass with the row statistics (m, 1/l, delta) handed over through a few LDS floats.
// LDS images are row-major [rows][64] bf16 with a 144-byte row stride: ds_read_b128 row reads are conflict-free and the
// transposed reads at most 2-way.  Padded rows are zero-filled, padded keys masked to probability 0.
#include "attention_mfma_common.h"
#include <stdlib.h>

namespace {

using namespace attn;

// ------------------------------------------------------------------------------------------------ forward
template <int NT, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const bf16_t* qkv, bf16_t* out, int S, int W, int H, int total_heads, float scale) {
    constexpr int KS = (NT + 1) / 2;
    constexpr bool ODD = (NT & 1) != 0;
    extern __shared__ __attribute__((aligned(16))) bf16_t lds_fwd[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int head = blockIdx.x * 4 + wave;
    if (head >= total_heads) return;
    bf16_t* Vs = lds_fwd + wave * (NT * 16 * LDR);
    const int b = head / H, h = head % H;
    const int64_t ld = 3 * (int64_t)W;
    const bf16_t* qb = qkv + (int64_t)b * S * ld + h * HD;
    const bf16_t* kb = qb + W;
    stage_head<NT>(Vs, qb + 2 * W, ld, S, lane);
    const int g = lane >> 4, c16 = lane & 15;

    bf16x8 Kf[NT][2];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) Kf[jt][ks] = row_frag_global(kb, ld, jt, ks, lane, S);
    bf16x8 Vf[4][KS];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int s = 0; s < KS; ++s)
            Vf[dt][s] = (ODD && s == KS - 1) ? tr_frag<false>(Vs, s, 16 * dt, lane) : tr_frag<true>(Vs, s, 16 * dt, lane);

    const int n_it = (S + 15) >> 4;
    for (int it = 0; it < n_it; ++it) {
        const bf16x8 q0 = row_frag_global(qb, ld, it, 0, lane, S), q1 = row_frag_global(qb, ld, it, 1, lane, S);
        const int i = it * 16 + c16;
        f32x4 sc[NT + 1];
        float m = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = MFMA16(Kf[jt][0], q0, a);
            a = MFMA16(Kf[jt][1], q1, a);
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
                const float p = __expf(sc[jt][r] - m);   // exp(-inf) = 0 for masked keys
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
            for (int dt = 0; dt < 4; ++dt) o[dt] = MFMA16(Vf[dt][s], pf, o[dt]);
        }
        if (i < S) {
            bf16_t* op = out + ((int64_t)b * S + i) * W + h * HD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) io<bf16_t>::st4(op + 16 * dt, o[dt] * inv);
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
// TWO waves per (batch, head), four heads per 512-thread workgroup: the two waves share the head's LDS images (Q, K, dO) and take
// alternate query tiles in pass 1 / alternate key tiles in pass 2, so a CU holds two waves per SIMD instead of one with the same
// LDS footprint - the one-wave form spent 72 % of its wave cycles in s_waitcnt (profiles/r02_attention_pmc_counters.txt) with
// nothing to switch to.  Workgroup barriers: after staging, between the passes (row statistics), before the column-sum hand-over.
template <int NT, bool CAUSAL, int WPH, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attn_bwd_mfma_kernel(const bf16_t* qkv, const bf16_t* d_out, bf16_t* d_qkv, int S, int W, int H, int total_heads,
                                                            float scale, float* cs_part /* [batch][3 W] column sums of d_qkv per image, or null */) {
    constexpr int KS = (NT + 1) / 2;
    constexpr bool ODD = (NT & 1) != 0;
    constexpr int IMG = NT * 16 * LDR;                 // elements per LDS image
    constexpr int HEAD_ELEMS = 3 * IMG + 3 * NT * 16 * 2;   // 3 images + 3 fp32 stat rows (2 bf16 slots per float)
    extern __shared__ __attribute__((aligned(16))) bf16_t lds_bwd[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = wave / WPH, half = wave % WPH;     // head slot of the workgroup, which of the head's WPH waves
    const int head_raw = blockIdx.x * (WAVES / WPH) + slot;
    const bool valid = head_raw < total_heads;           // a surplus slot repeats the last head's work and stores nothing (no early exit: barriers)
    const int head = valid ? head_raw : total_heads - 1;
    bf16_t* Ks = lds_bwd + slot * HEAD_ELEMS;
    bf16_t* Qs = Ks + IMG;
    bf16_t* Os = Qs + IMG;                              // dO
    float* st_m = (float*)(Os + IMG);                   // [NT*16] row max
    float* st_il = st_m + NT * 16;                      // 1 / row sum
    float* st_dl = st_il + NT * 16;                     // delta_i = sum_j p_ij dp_ij
    const int b = head / H, h = head % H;
    const int64_t ld = 3 * (int64_t)W;
    const bf16_t* qb = qkv + (int64_t)b * S * ld + h * HD;
    const bf16_t* vb = qb + 2 * W;
    const bf16_t* dob = d_out + (int64_t)b * S * W + h * HD;
    bf16_t* dqb = d_qkv + (int64_t)b * S * ld + h * HD;
    stage_head_part<NT, WPH>(Qs, qb, ld, S, lane, half);
    stage_head_part<NT, WPH>(Ks, qb + W, ld, S, lane, half);
    stage_head_part<NT, WPH>(Os, dob, W, S, lane, half);
    __syncthreads();
    const int g = lane >> 4, c16 = lane & 15;

    bf16x8 Vf[NT][2];     // V row fragments (rows = keys), used as A in pass 1 and as B in pass 2
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) Vf[jt][ks] = row_frag_global(vb, ld, jt, ks, lane, S);

    const int n_t = (S + 15) >> 4;
    f32x4 csq[4], csk[4], csv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) csq[dt] = csk[dt] = csv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---------------- pass 1: lane = query row i.  P, dS in registers -> dQ ; row statistics -> LDS
    for (int it = half; it < n_t; it += WPH) {
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
            d = MFMA16(Vf[jt][0], g0, d);
            d = MFMA16(Vf[jt][1], g1, d);
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
            for (int r = 0; r < 4; ++r) sc[jt][r] = sc[jt][r] * (dp[jt][r] - delta) * scale;   // dS (0 where p = 0)
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
        if (i < S && valid) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) io<bf16_t>::st4(dqb + (int64_t)i * ld + 16 * dt + 4 * g, dq[dt]);
        }
        if (cs_part) cs_add(csq, dq, i < S);
    }
    __syncthreads();   // the row statistics of BOTH waves' query tiles are in LDS
    // (statistics exist for the query tiles below n_t only: pass 2 skips the others)
    // ---------------- pass 2: lane = key row j.  P^T, dS^T products -> dV, dK
    for (int jt = half; jt < n_t; jt += WPH) {
        const bf16x8 k0 = row_frag_lds(Ks, jt, 0, lane), k1 = row_frag_lds(Ks, jt, 1, lane);
        const bf16x8 v0 = row_frag_global(vb, ld, jt, 0, lane, S), v1 = row_frag_global(vb, ld, jt, 1, lane, S);   // runtime jt: not Vf[jt] (scratch)
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
                if (ODD && it >= NT) continue;      // compile-time: the empty half of the last k-step
                if (it >= n_t) continue;            // query tile beyond the sequence (S <= 16 (NT - 1)): pass 1 wrote no statistics for it
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
        if (j < S && valid) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                io<bf16_t>::st4(dqb + (int64_t)j * ld + W + 16 * dt + 4 * g, dk[dt]);
                io<bf16_t>::st4(dqb + (int64_t)j * ld + 2 * W + 16 * dt + 4 * g, dv[dt]);
            }
        }
        if (cs_part) { cs_add(csk, dk, j < S); cs_add(csv, dv, j < S); }
    }
    if (cs_part) {   // the head's WPH waves are the only producers of its 192 columns for image b: waves 1.. hand their sums over, wave 0 adds (fixed order)
        cs_rows(csq); cs_rows(csk); cs_rows(csv);
        __syncthreads();                      // every wave is done reading the images: the K image becomes the hand-over buffer
        float* xch = (float*)Ks;              // [WPH - 1][3][4 dt][4 g][4 r] floats
        if (half != 0 && c16 == 0) {
            float* x = xch + (half - 1) * 192;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                *(f32x4*)(x + (0 * 16 + dt * 4 + g) * 4) = csq[dt];
                *(f32x4*)(x + (1 * 16 + dt * 4 + g) * 4) = csk[dt];
                *(f32x4*)(x + (2 * 16 + dt * 4 + g) * 4) = csv[dt];
            }
        }
        __syncthreads();
        if (half == 0 && c16 == 0 && valid) {
            float* dst = cs_part + (int64_t)b * 3 * W + h * HD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 q = csq[dt], k = csk[dt], v = csv[dt];
#pragma unroll
                for (int o = 0; o < WPH - 1; ++o) {
                    const float* x = xch + o * 192;
                    q += *(const f32x4*)(x + (0 * 16 + dt * 4 + g) * 4);
                    k += *(const f32x4*)(x + (1 * 16 + dt * 4 + g) * 4);
                    v += *(const f32x4*)(x + (2 * 16 + dt * 4 + g) * 4);
                }
                *(f32x4*)(dst + 16 * dt) = q;
                *(f32x4*)(dst + W + 16 * dt) = k;
                *(f32x4*)(dst + 2 * W + 16 * dt) = v;
            }
        }
    }
}

template <typename K>
int reserve_lds(K kernel, size_t bytes) {
    if (bytes > 65536) {
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return sc_set_error((int)e, "attention(mfma): cannot reserve %zu bytes of LDS: %s", bytes, hipGetErrorString(e));
    }
    return SC_OK;
}

template <int NT>
int launch_fwd(const bf16_t* qkv, bf16_t* out, int S, int W, int H, int total, bool causal, hipStream_t st) {
    const size_t lds = (size_t)4 * NT * 16 * LDR * sizeof(bf16_t);
    const dim3 grid((unsigned)sc_cdiv(total, 4));
    if (causal) {
        SC_TRY(reserve_lds(attn_fwd_mfma_kernel<NT, true>, lds));
        hipLaunchKernelGGL((attn_fwd_mfma_kernel<NT, true>), grid, dim3(256), lds, st, qkv, out, S, W, H, total, 0.125f);
    } else {
        SC_TRY(reserve_lds(attn_fwd_mfma_kernel<NT, false>, lds));
        hipLaunchKernelGGL((attn_fwd_mfma_kernel<NT, false>), grid, dim3(256), lds, st, qkv, out, S, W, H, total, 0.125f);
    }
    SC_CHECK_LAUNCH();
    return SC_OK;
}

// WPH = 2 (four heads per workgroup) measured 208 / 242 us at S = 77 / 50 against 250 / 276 us for WPH = 4 (profiles/r02_attention_times.txt)
// Heads per workgroup: four (512 threads, 116 / 146 KiB of LDS at S <= 64 / 80, one workgroup per CU) or two (256 threads, 58 / 73 KiB,
// two workgroups per CU, so one stages its images while the other computes).  Same waves per SIMD either way.  Stand-alone the
// two-head form is 1 % faster (245 / 208 against 248 / 210 us); inside the training step, next to the GEMMs of the other streams, the
// four-head form is the faster one (64.4 against 65.9 - 66.1 ms per step in one run): it is the default, SC_ATTN_BWD_WAVES=4 selects
// the two-head form (A/B).
template <int NT, int WPH, int WAVES>
int launch_bwd_w(const bf16_t* qkv, const bf16_t* d_out, bf16_t* d_qkv, int S, int W, int H, int total, bool causal, float* cs_part, hipStream_t st) {
    constexpr int HPW = WAVES / WPH;
    const size_t lds = (size_t)HPW * (3 * NT * 16 * LDR + 3 * NT * 16 * 2) * sizeof(bf16_t);
    const dim3 grid((unsigned)sc_cdiv(total, HPW));
    if (causal) {
        SC_TRY(reserve_lds(attn_bwd_mfma_kernel<NT, true, WPH, WAVES>, lds));
        hipLaunchKernelGGL((attn_bwd_mfma_kernel<NT, true, WPH, WAVES>), grid, dim3(WAVES * 64), lds, st, qkv, d_out, d_qkv, S, W, H, total, 0.125f, cs_part);
    } else {
        SC_TRY(reserve_lds(attn_bwd_mfma_kernel<NT, false, WPH, WAVES>, lds));
        hipLaunchKernelGGL((attn_bwd_mfma_kernel<NT, false, WPH, WAVES>), grid, dim3(WAVES * 64), lds, st, qkv, d_out, d_qkv, S, W, H, total, 0.125f, cs_part);
    }
    SC_CHECK_LAUNCH();
    return SC_OK;
}
template <int NT>
int launch_bwd(const bf16_t* qkv, const bf16_t* d_out, bf16_t* d_qkv, int S, int W, int H, int total, bool causal, float* cs_part, hipStream_t st) {
    static const bool four_heads = [] { const char* e = getenv("SC_ATTN_BWD_WAVES"); return !(e && e[0] == '4'); }();
    if (four_heads) return launch_bwd_w<NT, 2, 8>(qkv, d_out, d_qkv, S, W, H, total, causal, cs_part, st);
    return launch_bwd_w<NT, 2, 4>(qkv, d_out, d_qkv, S, W, H, total, causal, cs_part, st);
}

}  // namespace

// bf16, seq <= 80: MFMA path.  Returns SC_OK after launching, or 1 when the shape is not covered (caller falls back to the
// whole-head-in-LDS fp32-VALU kernel, which is also the fp32 parity path).
int sc_attention_mfma_fwd(const void* qkv, void* out, int64_t batch, int64_t seq, int64_t width, int64_t heads, int causal, hipStream_t st) {
    if (seq > 80) return 1;
    const int total = (int)(batch * heads);
    if (seq <= 64) return launch_fwd<4>((const bf16_t*)qkv, (bf16_t*)out, (int)seq, (int)width, (int)heads, total, causal != 0, st);
    return launch_fwd<5>((const bf16_t*)qkv, (bf16_t*)out, (int)seq, (int)width, (int)heads, total, causal != 0, st);
}
int sc_attention_mfma_bwd(const void* qkv, const void* d_out, void* d_qkv, int64_t batch, int64_t seq, int64_t width, int64_t heads, int causal,
                          float* cs_part, hipStream_t st) {
    if (seq > 80) return 1;
    const int total = (int)(batch * heads);
    if (seq <= 64) return launch_bwd<4>((const bf16_t*)qkv, (const bf16_t*)d_out, (bf16_t*)d_qkv, (int)seq, (int)width, (int)heads, total, causal != 0, cs_part, st);
    return launch_bwd<5>((const bf16_t*)qkv, (const bf16_t*)d_out, (bf16_t*)d_qkv, (int)seq, (int)width, (int)heads, total, causal != 0, cs_part, st);
}
