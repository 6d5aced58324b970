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
// LDS images are row-major [rows][64] bf16 in 128-byte rows with XOR-swizzled 16-byte chunks (attention_mfma_common.h): row reads and
// transposed reads are both conflict-free.  Padded rows are zero-filled, padded keys masked to probability 0.
#include "attention_mfma_common.h"
#include <stdlib.h>

namespace {

using namespace attn;

// ------------------------------------------------------------------------------------------------ forward
template <int NT, bool CAUSAL>
__global__ __launch_bounds__(256, NT >= 5 ? 3 : 4) void attn_fwd_mfma_kernel(const bf16_t* qkv, bf16_t* out, int S, int W, int H, int total_heads, float scale) {
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
    const float c = scale * 1.4426950408889634f;   // exp(x scale) = 2^(x c)
    // the next query tile's fragments are requested under this tile's arithmetic (a tile used to start with a round trip to global memory:
    // four or five per head, most of a head's time); unscaled scores (the maximum commutes with the positive scale), P = 2^((s - max) c) by
    // one fma and one v_exp, compare / select on the tiles that hold masked pairs only (the last key tile, the causal diagonal)
    bf16x8 q0 = row_frag_global(qb, ld, 0, 0, lane, S), q1 = row_frag_global(qb, ld, 0, 1, lane, S);
    for (int it = 0; it < n_it; ++it) {
        bf16x8 nq0 = q0, nq1 = q1;
        if (it + 1 < n_it) { nq0 = row_frag_global(qb, ld, it + 1, 0, lane, S); nq1 = row_frag_global(qb, ld, it + 1, 1, lane, S); }
        const int i = it * 16 + c16;
        f32x4 sc[NT + 1];
        float m = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = MFMA16(Kf[jt][0], q0, a);
            a = MFMA16(Kf[jt][1], q1, a);
            if (jt * 16 + 16 > S || (CAUSAL && jt >= it)) {   // wave-uniform: tiles with masked pairs (under the causal mask also the tiles past the diagonal)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = jt * 16 + 4 * g + r;
                    a[r] = (j < S && (!CAUSAL || j <= i)) ? a[r] : -INFINITY;
                }
            }
            m = fmaxf(fmaxf(m, fmaxf(a[0], a[1])), fmaxf(a[2], a[3]));
            sc[jt] = a;
        }
        sc[NT] = f32x4{0.f, 0.f, 0.f, 0.f};
        m = group_max(m);
        const float nmc = -m * c;
        float l = 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[jt][r], c, nmc));   // 2^(-inf) = 0 for masked keys
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
        q0 = nq0; q1 = nq1;
    }
}

// ------------------------------------------------------------------------------------------------ backward
// TWO waves per (batch, head), four heads per 512-thread workgroup: the two waves share the head's LDS images (Q, K, dO) and take
// alternate query tiles in pass 1 / alternate key tiles in pass 2, so a CU holds two waves per SIMD instead of one with the same
// LDS footprint - the one-wave form spent 72 % of its wave cycles in s_waitcnt (profiles/r02_attention_pmc_counters.txt) with
// nothing to switch to.  Workgroup barriers: after staging, between the passes (row statistics), before the column-sum hand-over.
//
// Round 3: the kernel is PERSISTENT and software-pipelined over head groups.  One workgroup per CU walks groups g, g + G, ...;
// the next group's images (12-15 16-byte pieces per lane) are requested into registers at the start of pass 2 of the
// current group and written to LDS behind its last barrier (its V fragments right after pass 2), so the global-load latency of a group (a third of a round before: every
// CU's workgroup started with 128 KiB of loads and nothing to do) hides under the previous group's arithmetic, and no workgroup
// launch / drain separates two groups.  The softmax arithmetic is cut down: exp2 with the scale folded into one FMA, one
// statistic e_i = log2(1 / l_i) - m_i c per query row (P = exp2(S c + e)), masks only on the tiles that contain a masked pair
// (wave-uniform branches), rows beyond the sequence get e = -inf (P = 0) instead of a per-element test.
template <int NT, int WPH>
struct BwdStage {
    static constexpr int NST = (2 * NT + WPH - 1) / WPH;
    uint4 q[NST], k[NST], o[NST];
};
template <int NT, int WPH, bool QK, bool O>
static __device__ __forceinline__ void bwd_stage_load(BwdStage<NT, WPH>& st, const bf16_t* qb, const bf16_t* dob, int64_t ld, int W, int S, int lane, int part) {
#pragma unroll
    for (int it2 = 0; it2 < BwdStage<NT, WPH>::NST; ++it2) {
        const int it = WPH * it2 + part;
        const int r = it * 8 + (lane >> 3), c = (lane & 7) * 8;
        const uint4 z = {0u, 0u, 0u, 0u};
        const bool live = it < 2 * NT && r < S;
        if (QK) {
            st.q[it2] = st.k[it2] = z;
            if (live) {
                st.q[it2] = *(const uint4*)(qb + (int64_t)r * ld + c);
                st.k[it2] = *(const uint4*)(qb + W + (int64_t)r * ld + c);
            }
        }
        if (O) {
            st.o[it2] = z;
            if (live) st.o[it2] = *(const uint4*)(dob + (int64_t)r * W + c);
        }
    }
}
template <int NT, int WPH>
static __device__ __forceinline__ void bwd_stage_store(const BwdStage<NT, WPH>& st, bf16_t* Qs, bf16_t* Ks, bf16_t* Os, int lane, int part) {
#pragma unroll
    for (int it2 = 0; it2 < BwdStage<NT, WPH>::NST; ++it2) {
        const int it = WPH * it2 + part;
        if (it >= 2 * NT) break;
        const int r = it * 8 + (lane >> 3), c = (lane & 7) * 8;
        *(uint4*)(Qs + img_off(r, c >> 3)) = st.q[it2];
        *(uint4*)(Ks + img_off(r, c >> 3)) = st.k[it2];
        *(uint4*)(Os + img_off(r, c >> 3)) = st.o[it2];
    }
}

template <int NT, bool CAUSAL, int WPH, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void attn_bwd_mfma_kernel(const bf16_t* qkv, const bf16_t* d_out, bf16_t* d_qkv, int S, int W, int H, int total_heads,
                                                            float scale, float* cs_part /* [batch][3 W] column sums of d_qkv per image, or null */) {
    constexpr int KS = (NT + 1) / 2;
    constexpr bool ODD = (NT & 1) != 0;
    constexpr int IMG = NT * 16 * LDR;                 // elements per LDS image
    constexpr int HEAD_ELEMS = 3 * IMG + 2 * NT * 16 * 2 + (3 * WPH - 2) * 64 * 2;   // 3 images + 2 fp32 stat rows + column-sum hand-over (2 bf16 slots per float)
    constexpr bool LATE_O = NT >= 5;   // registers: the dO pieces of the next group are requested after pass 2 (with its V fragments) instead of in front of it
    constexpr int HPW = WAVES / WPH;
    extern __shared__ __attribute__((aligned(16))) bf16_t lds_bwd[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = wave / WPH, half = wave % WPH;     // head slot of the workgroup, which of the head's WPH waves
    bf16_t* Ks = lds_bwd + slot * HEAD_ELEMS;
    bf16_t* Qs = Ks + IMG;
    bf16_t* Os = Qs + IMG;                              // dO
    float* st_e = (float*)(Os + IMG);                   // [NT*16] log2(1 / row sum) - row max * c   (-inf for rows beyond the sequence)
    float* st_dl = st_e + NT * 16;                      // -scale * delta_i,  delta_i = sum_j p_ij dp_ij
    float* csq_x = st_dl + NT * 16;                     // [WPH][4 dt][4 g][4 r] column sums of dq, row-reduced, parked here during pass 2 (registers)
    float* cskv_x = csq_x + WPH * 64;                   // [WPH - 1][2][64] column sums of dk, dv of the waves 1.. of the head
    const int64_t ld = 3 * (int64_t)W;
    const int g = lane >> 4, c16 = lane & 15;
    const int n_t = (S + 15) >> 4;
    const int n_groups = (total_heads + HPW - 1) / HPW;
    const float c = scale * 1.4426950408889634f;        // exp(x * scale) = exp2(x * c)

    int grp = blockIdx.x;
    if (grp >= n_groups) return;
    // a surplus slot of the last group repeats the last head's work and stores nothing (no early exit: barriers)
    int head_raw = grp * HPW + slot;
    BwdStage<NT, WPH> pre;
    bf16x8 Vf[NT][2];     // V row fragments (rows = keys): A operand in pass 1, B operand in pass 2
    {
        const int head = min(head_raw, total_heads - 1);
        const int b = head / H, h = head % H;
        const bf16_t* qb = qkv + (int64_t)b * S * ld + h * HD;
        bwd_stage_load<NT, WPH, true, true>(pre, qb, d_out + (int64_t)b * S * W + h * HD, ld, W, S, lane, half);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) Vf[jt][ks] = row_frag_global(qb + 2 * W, ld, jt, ks, lane, S);
    }
    for (;;) {
        const bool valid = head_raw < total_heads;
        const int head = valid ? head_raw : total_heads - 1;
        const int b = head / H, h = head % H;
        bf16_t* dqb = d_qkv + (int64_t)b * S * ld + h * HD;
        bwd_stage_store<NT, WPH>(pre, Qs, Ks, Os, lane, half);
        __syncthreads();

        f32x4 csq[4], csk[4], csv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) csq[dt] = csk[dt] = csv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // ---------------- pass 1: lane = query row i.  P, dS in registers -> dQ ; row statistics -> LDS
        // Tile -> wave: alternate for equal tiles; under the causal mask query tile t costs t + 1 key tiles (key tile t: n_t - t query tiles),
        // so the two waves of a head take the tiles in "snake" order from the heaviest (h l l h h l l ...): 8 against 7 units at five
        // tiles, where alternating gave 9 against 6 and the lighter wave waited a third of each pass at the barrier
        for (int it = 0; it < n_t; ++it) {
            if (CAUSAL && WPH == 2 ? ((((n_t - it) >> 1) & 1) != half) : (it % WPH != half)) continue;
            const bf16x8 q0 = row_frag_lds(Qs, it, 0, lane), q1 = row_frag_lds(Qs, it, 1, lane);
            const bf16x8 g0 = row_frag_lds(Os, it, 0, lane), g1 = row_frag_lds(Os, it, 1, lane);
            const int i = it * 16 + c16;
            f32x4 sc[NT + 1], dp[NT];
            float m = -INFINITY;
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                sc[jt] = dp[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
                if ((CAUSAL && jt > it) || jt >= n_t) continue;   // key tile in the future of this query tile / beyond the sequence: p = dS = 0 (wave-uniform)
                f32x4 a = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};
                a = MFMA16(row_frag_lds(Ks, jt, 0, lane), q0, a);
                a = MFMA16(row_frag_lds(Ks, jt, 1, lane), q1, a);
                d = MFMA16(Vf[jt][0], g0, d);
                d = MFMA16(Vf[jt][1], g1, d);
                if (jt * 16 + 16 > S || (CAUSAL && jt == it)) {   // the only tiles with masked pairs (wave-uniform)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = jt * 16 + 4 * g + r;
                        const bool ok = j < S && (!CAUSAL || j <= i);
                        a[r] = ok ? a[r] : -INFINITY;
                    }
                }
                m = fmaxf(fmaxf(m, fmaxf(a[0], a[1])), fmaxf(a[2], a[3]));
                sc[jt] = a;
                dp[jt] = d;
            }
            sc[NT] = f32x4{0.f, 0.f, 0.f, 0.f};
            m = group_max(m);            // raw scores: scale > 0, so this is the row of the maximal scaled score too; key 0 is never masked
            const float mc = m * c;
            float l = 0.f, dsum = 0.f;
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                if ((CAUSAL && jt > it) || jt >= n_t) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[jt][r], c, -mc));   // exp2(-inf) = 0 for masked keys
                    sc[jt][r] = p;
                    l += p;
                    dsum = __builtin_fmaf(p, dp[jt][r], dsum);
                }
            }
            l = group_sum(l);
            dsum = group_sum(dsum);
            const float inv = 1.0f / l;
            const float delta = dsum * inv, c2 = inv * scale, dc2 = -delta * c2;
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                if ((CAUSAL && jt > it) || jt >= n_t) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[jt][r] *= __builtin_fmaf(dp[jt][r], c2, dc2);   // dS = p (dp - delta) / l * scale (0 where p = 0: dp, delta are finite)
            }
            if (g == 0) {
                const bool live = i < S;
                st_e[i] = live ? __builtin_amdgcn_logf(inv) - mc : -INFINITY;
                st_dl[i] = live ? -delta * scale : 0.f;
            }
            f32x4 dq[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if ((CAUSAL && 2 * s > it) || 2 * s >= n_t) continue;   // both key tiles of this k-step are masked out
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
        if (cs_part) {
            cs_rows(csq);
            if (c16 == 0) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) *(f32x4*)(csq_x + half * 64 + (dt * 4 + g) * 4) = csq[dt];
            }
        }
        // pass 2's tiles of this wave and the V fragments of the first one (requested in front of the barrier: they land while the wave waits)
        auto mine2 = [&](int jt) { return CAUSAL && WPH == 2 ? ((((jt + 1) >> 1) & 1) == half) : (jt % WPH == half); };
        auto next2 = [&](int jt) { do ++jt; while (jt < n_t && !mine2(jt)); return jt; };
        const bf16_t* vb = qkv + (int64_t)b * S * ld + h * HD + 2 * W;
        __syncthreads();   // the row statistics of BOTH waves' query tiles are in LDS
        // (statistics exist for the query tiles below n_t only: pass 2 skips the others)

        // the next group's operands: requested here, consumed behind this group's last barrier
        const int nxt = grp + (int)gridDim.x;
        const bool more = nxt < n_groups;
        const int nhead_raw = nxt * HPW + slot;
        const int nh = min(nhead_raw, total_heads - 1);
        const bf16_t* nq = qkv + (int64_t)(nh / H) * S * ld + (nh % H) * HD;
        const bf16_t* ndo = d_out + (int64_t)(nh / H) * S * W + (nh % H) * HD;
        int jt = next2(-1);
        bf16x8 v0, v1;
        if (jt < n_t) { v0 = row_frag_global(vb, ld, jt, 0, lane, S); v1 = row_frag_global(vb, ld, jt, 1, lane, S); }   // runtime jt: not Vf[jt] (scratch)
        if (more) bwd_stage_load<NT, WPH, true, !LATE_O>(pre, nq, ndo, ld, W, S, lane, half);
        // ---------------- pass 2: lane = key row j.  P^T, dS^T products -> dV, dK
        while (jt < n_t) {
            const int jn = next2(jt);
            bf16x8 n0 = v0, n1 = v1;
            if (jn < n_t) { n0 = row_frag_global(vb, ld, jn, 0, lane, S); n1 = row_frag_global(vb, ld, jn, 1, lane, S); }   // one key tile ahead
            const bf16x8 k0 = row_frag_lds(Ks, jt, 0, lane), k1 = row_frag_lds(Ks, jt, 1, lane);
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
                    const f32x4 ee = *(const f32x4*)(st_e + it * 16 + 4 * g), dl = *(const f32x4*)(st_dl + it * 16 + 4 * g);
                    // rows beyond the sequence carry e = -inf (p = 0, and their dO rows are zero); keys beyond it only produce lanes that are
                    // never stored; so the only pairs to mask are those above the diagonal of the diagonal tile
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(a[r], c, ee[r]));
                        if (CAUSAL && it == jt) p = (j <= it * 16 + 4 * g + r) ? p : 0.f;
                        pt[u][r] = p;
                        dst[u][r] = p * __builtin_fmaf(d[r], scale, dl[r]);
                    }
                }
                if (CAUSAL && 2 * s + 1 < jt) continue;
                if (2 * s >= n_t) continue;
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
            v0 = n0; v1 = n1; jt = jn;
        }
        // the head's WPH waves are the only producers of its 192 columns for image b: waves 1.. hand their sums over through LDS words of their
        // own, wave 0 adds in a fixed order behind the group's last barrier (which also frees the images for the next group's staging)
        if (cs_part) {
            cs_rows(csk); cs_rows(csv);
            if (half != 0 && c16 == 0) {
                float* x = cskv_x + (half - 1) * 128;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    *(f32x4*)(x + (dt * 4 + g) * 4) = csk[dt];
                    *(f32x4*)(x + 64 + (dt * 4 + g) * 4) = csv[dt];
                }
            }
        }
        __syncthreads();   // every wave is done with this group's images; the partial column sums are in LDS
        if (cs_part && half == 0 && c16 == 0 && valid) {
            float* dst = cs_part + (int64_t)b * 3 * W + h * HD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 q = *(const f32x4*)(csq_x + (dt * 4 + g) * 4), k = csk[dt], v = csv[dt];
#pragma unroll
                for (int o = 0; o < WPH - 1; ++o) {
                    q += *(const f32x4*)(csq_x + (o + 1) * 64 + (dt * 4 + g) * 4);
                    k += *(const f32x4*)(cskv_x + o * 128 + (dt * 4 + g) * 4);
                    v += *(const f32x4*)(cskv_x + o * 128 + 64 + (dt * 4 + g) * 4);
                }
                *(f32x4*)(dst + 16 * dt) = q;
                *(f32x4*)(dst + W + 16 * dt) = k;
                *(f32x4*)(dst + 2 * W + 16 * dt) = v;
            }
        }
        if (!more) break;
        if (LATE_O) bwd_stage_load<NT, WPH, false, true>(pre, nq, ndo, ld, W, S, lane, half);
#pragma unroll
        for (int jt2 = 0; jt2 < NT; ++jt2)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) Vf[jt2][ks] = row_frag_global(nq + 2 * W, ld, jt2, ks, lane, S);
        grp = nxt;
        head_raw = nhead_raw;
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

// Two waves per head, four heads per 512-thread workgroup (100 / 125 KiB of LDS at S <= 64 / 80, one persistent workgroup per CU).
// Round 3, stand-alone at the step's shapes (tools/attn_bench.py; profiles/r03_attention_times.txt): 246 / 210 us (round 2) -> 177 / 161 us
// at S = 50 / 77.  Measured on the way and dropped: workgroups of two heads (two per CU) or of one head (four per CU, so that a barrier
// only joins the two waves of a head): 175 / 171 and 175 / 186 us; three waves per SIMD without the register prefetch: 194 / 230 us;
// four waves per head (round 2): 250 / 276 us.  Conflict-free LDS images alone changed nothing: the kernel waits on dependency chains
// (LDS read -> MFMA -> softmax arithmetic -> MFMA), not on LDS bandwidth (LDS active 14 % of the time, profiles/r03_attention_pmc_counters.txt).
template <int NT, int WPH, int WAVES>
int launch_bwd_w(const bf16_t* qkv, const bf16_t* d_out, bf16_t* d_qkv, int S, int W, int H, int total, bool causal, float* cs_part, hipStream_t st) {
    constexpr int HPW = WAVES / WPH;
    const size_t lds = (size_t)HPW * (3 * NT * 16 * LDR + 2 * NT * 16 * 2 + (3 * WPH - 2) * 64 * 2) * sizeof(bf16_t);
    // persistent: as many workgroups as the chip holds at once (LDS-bound), each walking groups g, g + G, ...
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    int64_t per_cu = (int64_t)(160 * 1024) / (int64_t)lds > 0 ? (int64_t)(160 * 1024) / (int64_t)lds : 1;
    if (per_cu * WAVES > 8) per_cu = 8 / WAVES;   // registers: two waves per SIMD
    const int64_t groups = sc_cdiv(total, HPW), resident = (int64_t)cus * per_cu;
    const dim3 grid((unsigned)(groups < resident ? groups : resident));
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
    return launch_bwd_w<NT, 2, 8>(qkv, d_out, d_qkv, S, W, H, total, causal, cs_part, st);
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
