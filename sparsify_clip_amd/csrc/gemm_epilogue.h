// Fused GEMM epilogue shared by the fp32 and bf16 MFMA kernels.
//   v = alpha*acc (+ bias[n]) ; pre_out = v ; v = act(v) ; v *= gelu'(dgelu_pre) ; v += resid ; v += beta*C_old
// TAux is the activation dtype of pre_out / dgelu_pre (float in the fp32 GEMM, bf16 in the bf16 GEMM).
#pragma once
#include "common.h"

struct EpiParams {
    float alpha, beta;
    const float* bias;
    void* pre_out;
    int act;
    int resid_dtype;
    const void* resid;
    const void* dgelu_pre;
    int64_t ld_aux;
    float* colsum;            // [N] column sums of the stored output (host side: where they go)
    void* colsum_ws;
    size_t colsum_ws_bytes;
    int colsum_accumulate;
    float* cs_partial;        // device side: [2 * tiles_m][N] partial sums written by the 256x256 NT kernel, or null
    unsigned* tickets;        // persistent NT kernel: caller-owned tile tickets (16 zeroed words, one stream at a time) or null = static tile lists
    // internal mode 1: w = (m==n) ? 0 : exp(-t * max(rowv[m] + colv[n] - 2*acc, 0))   (pairwise-distance kernel of lunif)
    int mode;
    const float* rowv;
    const float* colv;
    float t;
};

static inline EpiParams epi_plain(float alpha = 1.f, float beta = 0.f) {
    EpiParams e;
    e.alpha = alpha; e.beta = beta; e.bias = nullptr; e.pre_out = nullptr; e.act = 0; e.resid_dtype = SC_F32;
    e.resid = nullptr; e.dgelu_pre = nullptr; e.ld_aux = 0; e.mode = 0; e.rowv = nullptr; e.colv = nullptr; e.t = 0.f;
    e.colsum = nullptr; e.colsum_ws = nullptr; e.colsum_ws_bytes = 0; e.colsum_accumulate = 0; e.cs_partial = nullptr; e.tickets = nullptr;
    return e;
}

static inline int epi_from_abi(const sc_gemm_epilogue* a, int /*aux_dtype*/, EpiParams& e) {
    e = epi_plain();
    if (!a) return SC_OK;
    e.alpha = a->alpha; e.beta = a->beta; e.bias = a->bias; e.pre_out = a->pre_out; e.act = a->act;
    e.resid_dtype = a->resid_dtype; e.resid = a->resid; e.dgelu_pre = a->dgelu_pre; e.ld_aux = a->ld_aux;
    e.colsum = a->colsum; e.colsum_ws = a->colsum_ws; e.colsum_ws_bytes = (size_t)a->colsum_ws_bytes; e.colsum_accumulate = a->colsum_accumulate;
    e.tickets = (unsigned*)a->tile_tickets;
    if (e.tickets && ((size_t)e.tickets & 63)) return sc_set_error(SC_ERR_ALIGN, "epilogue: tile_tickets must be 64-byte aligned");
    if (e.colsum && !e.colsum_ws) return sc_set_error(SC_ERR_WORKSPACE, "epilogue: colsum needs colsum_ws");
    if (e.act != 0 && e.act != 1) return sc_set_error(SC_ERR_ARG, "epilogue: unknown activation %d", e.act);
    if (e.resid && e.resid_dtype != SC_F32 && e.resid_dtype != SC_BF16) return sc_set_error(SC_ERR_DTYPE, "epilogue: bad resid dtype");
    if ((e.pre_out || e.resid || e.dgelu_pre) && e.ld_aux <= 0) return sc_set_error(SC_ERR_SHAPE, "epilogue: ld_aux missing");
    return SC_OK;
}

#ifdef __HIPCC__
template <typename TAux>
__device__ __forceinline__ float epi_scalar(const EpiParams& e, float acc, int m, int n, const float* c_old) {
    if (e.mode == 1) return (m == n) ? 0.f : expf(-e.t * fmaxf(e.rowv[m] + e.colv[n] - 2.f * acc, 0.f));
    float v = e.alpha * acc;
    if (e.bias) v += e.bias[n];
    const int64_t off = (int64_t)m * e.ld_aux + n;
    if (e.pre_out) io<TAux>::st((TAux*)e.pre_out + off, v);
    if (e.act == 1) v = gelu_f(v);
    if (e.dgelu_pre) v *= gelu_grad_f(io<TAux>::ld((const TAux*)e.dgelu_pre + off));
    if (e.resid) v += (e.resid_dtype == SC_F32) ? ((const float*)e.resid)[off] : bf16_to_f32(((const bf16_t*)e.resid)[off]);
    if (e.beta != 0.f) v += e.beta * (*c_old);
    return v;
}

// four consecutive n at one m (n % 4 == 0, ld_aux % 4 == 0)
template <typename TAux>
__device__ __forceinline__ f32x4 epi_vec4(const EpiParams& e, f32x4 acc, int m, int n, const float* c_old /*fp32 C or null*/) {
    f32x4 v = acc * e.alpha;
    if (e.bias) v += *(const f32x4*)(e.bias + n);
    const int64_t off = (int64_t)m * e.ld_aux + n;
    if (e.pre_out) io<TAux>::st4((TAux*)e.pre_out + off, v);
    constexpr bool FAST = sizeof(TAux) == 2;   // bf16 activations: the 1e-5 polynomial forms (common.h) are exact at bf16 resolution
    if (e.act == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = FAST ? gelu_fast(v[j]) : gelu_f(v[j]);
    }
    if (e.dgelu_pre) {
        const f32x4 h = io<TAux>::ld4((const TAux*)e.dgelu_pre + off);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= FAST ? gelu_grad_fast(h[j]) : gelu_grad_f(h[j]);
    }
    if (e.resid) v += (e.resid_dtype == SC_F32) ? io<float>::ld4((const float*)e.resid + off) : io<bf16_t>::ld4((const bf16_t*)e.resid + off);
    if (e.beta != 0.f && c_old) v += *(const f32x4*)c_old * e.beta;
    return v;
}
#endif
