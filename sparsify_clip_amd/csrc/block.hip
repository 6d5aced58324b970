// One pre-LN residual attention block of the CLIP towers (open_clip ResidualAttentionBlock):
//     x_mid = x_in  + out_proj(attention(in_proj(ln_1(x_in))))
//     x_out = x_mid + c_proj(gelu(c_fc(ln_2(x_mid))))
// forward and backward as a fixed sequence of launches on the caller's stream.  The residual stream and its
// gradient stay fp32; GEMM operands are `dtype` (bf16 -> MFMA 16x16x32, fp32 -> MFMA 32x32x2).  Bias, exact-erf
// GELU, GELU', and the residual adds are fused into the GEMM epilogues; LayerNorm backward also emits the
// `dtype` copy of the residual gradient that the next GEMM consumes.
#include "common.h"
#include <stdlib.h>
#include "gemm_epilogue.h"

int sc_gemm_f32_launch(int trans_a, int trans_b, int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, const float* b, int64_t ldb,
                       float* c, int64_t ldc, const EpiParams& epi, hipStream_t stream);
int sc_gemm_bf16_nt_launch(int64_t m, int64_t n, int64_t k, const void* a, int64_t lda, const void* b, int64_t ldb, void* c, int64_t ldc,
                           int out_dtype, const EpiParams& epi, hipStream_t stream);
int sc_gemm_bf16_tn_launch(int64_t m, int64_t n, int64_t r, const void* a, int64_t lda, const void* b, int64_t ldb, float* c, int64_t ldc,
                           float alpha, float beta, void* ws, size_t ws_bytes, hipStream_t stream, float* colsum_a, float colsum_beta, int conv_w = 0);
size_t sc_gemm_bf16_tn_ws(int64_t m, int64_t n, int64_t r);
size_t sc_gemm_bf16_tn_group_ws(int nprob, const int64_t* m, const int64_t* n, int64_t r);
int sc_gemm_bf16_tn_group_launch(int nprob, const int64_t* m, const int64_t* n, int64_t r, const void* const* a, const int64_t* lda, const void* const* b,
                                 const int64_t* ldb, float* const* c, const int64_t* ldc, float alpha, float beta, void* ws, size_t ws_bytes,
                                 hipStream_t stream);

int sc_attention_f32_composed_fwd(const float* qkv, float* out, int64_t batch, int64_t seq, int64_t width, int64_t heads, int causal, void* ws,
                                  size_t ws_bytes, hipStream_t st);
int sc_attention_f32_composed_bwd(const float* qkv, const float* d_out, float* d_qkv, int64_t batch, int64_t seq, int64_t width, int64_t heads, int causal,
                                  void* ws, size_t ws_bytes, hipStream_t st);

bool sc_attention_f32_bwd_fits_lds(int64_t seq);

int sc_gelu_fwd_bf16(const void* pre, void* act, int64_t n_elems, hipStream_t st);
int sc_dgelu_mul_colsum_bf16(void* dh, const void* pre, int64_t rows, int64_t n, float* colsum, int accumulate, void* ws, size_t ws_bytes, hipStream_t st);

namespace {
// SC_BLOCK_UNFUSE_GELU=1: GELU / GELU' as separate HBM-bound kernels instead of GEMM epilogues (A/B: the step is GEMM-bound and other
// kernels run in the GEMMs' shadow, so moving epilogue work out of the GEMMs might pay; bf16 path only)
bool unfuse_gelu() {
    static const bool on = [] { const char* e = sc_debug_env("SC_BLOCK_UNFUSE_GELU"); return e && e[0] == '1'; }();
    return on;
}

// y[rows,n] = x[rows,k] W[n,k]^T (+ epilogue); W in torch [out,in] layout
int linear_fwd(int dtype, int64_t rows, int64_t n, int64_t k, const void* x, const void* w, void* y, int out_dtype, const EpiParams& epi,
               hipStream_t st, void* tickets) {
    if (dtype == SC_BF16) {
        EpiParams e = epi;
        e.tickets = (unsigned*)tickets;
        return sc_gemm_bf16_nt_launch(rows, n, k, x, k, w, k, y, n, out_dtype, e, st);
    }
    return sc_gemm_f32_launch(0, 1, rows, n, k, (const float*)x, k, (const float*)w, k, (float*)y, n, epi, st);
}
// dx[rows,k] = dy[rows,n] W[n,k]  (bf16: NT against the [in,out] copy wt[k,n])
int linear_dx(int dtype, int64_t rows, int64_t n, int64_t k, const void* dy, const void* w, const void* wt, void* dx, const EpiParams& epi,
              hipStream_t st, void* tickets) {
    if (dtype == SC_BF16) {
        EpiParams e = epi;
        e.tickets = (unsigned*)tickets;
        return sc_gemm_bf16_nt_launch(rows, k, n, dy, n, wt, n, dx, k, SC_BF16, e, st);
    }
    return sc_gemm_f32_launch(0, 0, rows, k, n, (const float*)dy, n, (const float*)w, k, (float*)dx, k, epi, st);
}
// dw[n,k] (+)= dy[rows,n]^T x[rows,k]; bf16 with db != null: db[n] (+)= column sums of dy, fused into the same kernel (the
// weight-gradient GEMM stages every dy tile in LDS anyway, so the bias gradient costs no extra pass over dy)
int linear_dw(int dtype, int64_t rows, int64_t n, int64_t k, const void* dy, const void* x, float* dw, int accumulate, void* ws, size_t ws_bytes,
              hipStream_t st, float* db = nullptr) {
    const float beta = accumulate ? 1.f : 0.f;
    if (dtype == SC_BF16) return sc_gemm_bf16_tn_launch(n, k, rows, dy, n, x, k, dw, k, 1.f, beta, ws, ws_bytes, st, db, beta);
    return sc_gemm_f32_launch(1, 0, n, k, rows, (const float*)dy, n, (const float*)x, k, dw, k, epi_plain(1.f, beta), st);
}

int check_desc(const sc_block_desc* d, const char* who) {
    SC_REQUIRE(d != nullptr, SC_ERR_ARG, "%s: null descriptor", who);
    SC_REQUIRE(d->dtype == SC_BF16 || d->dtype == SC_F32, SC_ERR_DTYPE, "%s: bad dtype %d", who, d->dtype);
    SC_REQUIRE(d->batch > 0 && d->seq > 0 && d->width > 0 && d->heads > 0 && d->mlp_width > 0, SC_ERR_SHAPE, "%s: bad dims", who);
    SC_REQUIRE(d->width == d->heads * 64, SC_ERR_SHAPE, "%s: width must be heads*64", who);
    SC_REQUIRE(d->width % 64 == 0 && d->mlp_width % 64 == 0, SC_ERR_SHAPE, "%s: widths must be multiples of 64", who);
    SC_REQUIRE(d->x_in && d->x_mid && d->x_out && d->ln1_out && d->qkv && d->attn_out && d->ln2_out && d->h_pre && d->h_act && d->ln1_mean &&
                   d->ln1_rstd && d->ln2_mean && d->ln2_rstd,
               SC_ERR_ARG, "%s: null activation buffer", who);
    SC_REQUIRE(d->w_qkv && d->w_o && d->w_fc1 && d->w_fc2 && d->ln1_g && d->ln1_b && d->ln2_g && d->ln2_b && d->b_qkv && d->b_o && d->b_fc1 && d->b_fc2,
               SC_ERR_ARG, "%s: null parameter", who);
    return SC_OK;
}

}  // namespace

namespace {
// byte offsets of the partial-sum regions of a backward block inside d->ws (deferred reductions: every producer keeps its partials until
// the block's one reduction launch): c_fc bias gradient (GEMM epilogue: one row per 64 output rows + 2, at least the 1024 rows the GEMM asks
// for), ln_2 (at most 1024 rows of three sums), in_proj bias gradient (attention backward: one row per image), ln_1
struct ReduceRegions { size_t fc1, ln2, attn, ln1, total; };
ReduceRegions reduce_regions(int64_t rows, int64_t batch, int64_t width, int64_t mlp_width) {
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    ReduceRegions r;
    const size_t ln = up((size_t)1024 * 3 * width * sizeof(float));
    size_t fc1 = up((size_t)(rows / 64 + 2) * mlp_width * sizeof(float));
    const size_t fc1_min = up((size_t)1024 * mlp_width * sizeof(float));
    if (fc1 < fc1_min) fc1 = fc1_min;
    r.fc1 = 0;
    r.ln2 = fc1;
    r.attn = r.ln2 + ln;
    r.ln1 = r.attn + up((size_t)(batch > 1024 ? batch : 1024) * 3 * width * sizeof(float));   // one row per image; 1024: what the un-fused column-sum pass may ask for
    r.total = r.ln1 + ln;
    return r;
}
}  // namespace

extern "C" size_t sc_block_workspace_bytes(int64_t rows, int64_t width, int64_t mlp_width, int dtype) {
    if (rows <= 0 || width <= 0 || mlp_width <= 0) return 0;
    size_t need = (size_t)1024 * 3 * (size_t)width * sizeof(float);                       // LayerNorm partials
    const size_t widest = (size_t)(mlp_width > 3 * width ? mlp_width : 3 * width);
    need = need > 1024 * widest * sizeof(float) ? need : 1024 * widest * sizeof(float);   // column-sum partials
    if (dtype == SC_BF16) {
        const int64_t shapes[4][2] = {{3 * width, width}, {width, width}, {mlp_width, width}, {width, mlp_width}};
        for (auto& s : shapes) {
            const size_t w = sc_gemm_bf16_tn_ws(s[0], s[1], rows);
            need = need > w ? need : w;
        }
        if (rows % 64 == 0) {   // the four weight gradients as one grouped launch
            const int64_t gm[4] = {width, mlp_width, width, 3 * width}, gn[4] = {mlp_width, width, width, width};
            const size_t w = sc_gemm_bf16_tn_group_ws(4, gm, gn, rows);
            need = need > w ? need : w;
        }
    }
    if (dtype == SC_BF16) {   // the four column-sum producers of a backward block keep their partials side by side until ONE reduction launch
        const size_t w = reduce_regions(rows, sc_cdiv(rows, 16), width, mlp_width).total;   // sequences of 16 positions or more keep the regions apart
        need = need > w ? need : w;
    }
    return ((need + 255) / 256) * 256;
}

extern "C" int sc_block_fwd(const sc_block_desc* d, void* stream) {
    SC_TRY(check_desc(d, "sc_block_fwd"));
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = d->batch * d->seq, W = d->width, MLP = d->mlp_width;
    const int dt = d->dtype;
    // ln_1 -> in_proj (+bias)
    SC_TRY(sc_layernorm_fwd(d->x_in, rows, W, d->ln1_g, d->ln1_b, d->ln1_out, dt, d->ln1_mean, d->ln1_rstd, stream));
    EpiParams e = epi_plain();
    e.bias = d->b_qkv;
    SC_TRY(linear_fwd(dt, rows, 3 * W, W, d->ln1_out, d->w_qkv, d->qkv, dt, e, st, d->tile_tickets));
    if (dt == SC_F32 && d->seq > 128)   // fp32 parity path of long sequences: composed from the fp32 GEMM, needs a workspace
        SC_TRY(sc_attention_f32_composed_fwd((const float*)d->qkv, (float*)d->attn_out, d->batch, d->seq, W, d->heads, d->causal, d->ws, d->ws_bytes, (hipStream_t)stream));
    else
        SC_TRY(sc_attention_fwd_stats(d->qkv, d->attn_out, d->attn_lse, dt, d->batch, d->seq, W, d->heads, d->causal, stream));   // lse NULL / shapes without statistics: the plain forward
    // out_proj (+bias) + residual -> fp32 x_mid
    e = epi_plain();
    e.bias = d->b_o; e.resid = d->x_in; e.resid_dtype = SC_F32; e.ld_aux = W;
    SC_TRY(linear_fwd(dt, rows, W, W, d->attn_out, d->w_o, d->x_mid, SC_F32, e, st, d->tile_tickets));
    // ln_2 -> c_fc (+bias, keep pre-activation) -> GELU
    SC_TRY(sc_layernorm_fwd(d->x_mid, rows, W, d->ln2_g, d->ln2_b, d->ln2_out, dt, d->ln2_mean, d->ln2_rstd, stream));
    e = epi_plain();
    if (dt == SC_BF16 && unfuse_gelu()) {
        e.bias = d->b_fc1;
        SC_TRY(linear_fwd(dt, rows, MLP, W, d->ln2_out, d->w_fc1, d->h_pre, dt, e, st, d->tile_tickets));
        SC_TRY(sc_gelu_fwd_bf16(d->h_pre, d->h_act, rows * MLP, st));
    } else {
        e.bias = d->b_fc1; e.pre_out = d->h_pre; e.act = 1; e.ld_aux = MLP;
        SC_TRY(linear_fwd(dt, rows, MLP, W, d->ln2_out, d->w_fc1, d->h_act, dt, e, st, d->tile_tickets));
    }
    // c_proj (+bias) + residual -> fp32 x_out
    e = epi_plain();
    e.bias = d->b_fc2; e.resid = d->x_mid; e.resid_dtype = SC_F32; e.ld_aux = W;
    SC_TRY(linear_fwd(dt, rows, W, MLP, d->h_act, d->w_fc2, d->x_out, SC_F32, e, st, d->tile_tickets));
    return SC_OK;
}

namespace {

int block_bwd_impl(const sc_block_desc* d, const float* dx_out, const void* dx_out_t, float* dx_in, void* dx_in_t, hipStream_t st, hipStream_t side) {
    SC_TRY(check_desc(d, "sc_block_bwd"));
    SC_REQUIRE(dx_out && dx_in, SC_ERR_ARG, "sc_block_bwd: null gradient buffer");
    SC_REQUIRE(d->d_h && d->d_ln && d->d_qkv && d->d_attn && d->dx_mid && d->ws, SC_ERR_ARG, "sc_block_bwd: null scratch buffer");
    SC_REQUIRE(d->g_ln1_g && d->g_ln1_b && d->g_w_qkv && d->g_b_qkv && d->g_w_o && d->g_b_o && d->g_ln2_g && d->g_ln2_b && d->g_w_fc1 && d->g_b_fc1 &&
                   d->g_w_fc2 && d->g_b_fc2,
               SC_ERR_ARG, "sc_block_bwd: null gradient output");
    const int64_t rows = d->batch * d->seq, W = d->width, MLP = d->mlp_width;
    const int dt = d->dtype, acc = d->accumulate;
    const bool bf = dt == SC_BF16;
    if (bf) SC_REQUIRE(d->wt_qkv && d->wt_o && d->wt_fc1 && d->wt_fc2 && d->d_res_t, SC_ERR_ARG, "sc_block_bwd: bf16 needs the [in,out] weight copies and d_res_t");
    const size_t need = sc_block_workspace_bytes(rows, W, MLP, dt);
    SC_REQUIRE(d->ws_bytes >= need, SC_ERR_WORKSPACE, "sc_block_bwd: workspace too small");
    // fp32 path: the GEMM operand of the weight gradients is dx_out itself, which the caller may overwrite in place -> one stream
    const bool two = side != nullptr && side != st && bf;
    if (two) SC_REQUIRE(d->ws_side && d->ws_side_bytes >= need, SC_ERR_WORKSPACE, "sc_block_bwd_async: ws_side missing or too small");
    if (two) SC_REQUIRE(d->events[0] && d->events[1] && d->events[2] && d->events[3], SC_ERR_ARG, "sc_block_bwd_async: d->events[0..3] must be caller-owned HIP events (sc_event_create)");
    hipStream_t ss = two ? side : st;                 // stream of the weight-gradient work
    void* wsw = two ? d->ws_side : d->ws;
    const size_t wsw_bytes = two ? d->ws_side_bytes : d->ws_bytes;
    void* stream = (void*)st;
    int nev = 0;
    // "everything enqueued on st so far is visible to the side stream"
    auto publish = [&]() -> int {
        if (!two) return SC_OK;
        hipEvent_t e = (hipEvent_t)d->events[nev++ & 3];
        hipError_t rc = hipEventRecord(e, st);
        if (rc == hipSuccess) rc = hipStreamWaitEvent(ss, e, 0);
        return rc == hipSuccess ? SC_OK : sc_set_error((int)rc, "sc_block_bwd_async: event: %s", hipGetErrorString(rc));
    };

    // GEMM-operand view of dx_out
    const void* g = dx_out;
    if (bf) {
        if (dx_out_t) g = dx_out_t;
        else {
            SC_TRY(sc_cast_f32_to_bf16(dx_out, d->d_res_t, rows * W, stream));
            g = d->d_res_t;
        }
    }
    // With a side stream the three weight-gradient GEMMs whose operands exist before the attention backward are held back until
    // the main stream gets there (SC_BLOCK_DW_GATE=0: released as early as possible): GEMM kernels fill every CU and run one
    // after the other whichever stream they come from, so issuing them early only makes them queue between the
    // activation-gradient GEMMs, while the attention backward - latency bound, a few waves per CU - otherwise leaves the
    // MFMA units idle for its whole duration.
    static const bool gate_env = [] { const char* e = sc_debug_env("SC_BLOCK_DW_GATE"); return !(e && e[0] == '0'); }();
    const bool gate = two && gate_env;
    // SC_BLOCK_DW_GROUP=0: four separate weight-gradient launches (A/B); default: ONE grouped launch per block, after the attention
    // backward, when the bias gradients are taken elsewhere (fcs) and the rows are whole K-tiles
    static const bool group_env = [] { const char* e = sc_debug_env("SC_BLOCK_DW_GROUP"); return !(e && e[0] == '0'); }();
    const bool fcs_early = bf && [] { const char* e = sc_debug_env("SC_BLOCK_FUSE_CS"); return !(e && e[0] == '0'); }();
    const bool grouped = group_env && fcs_early && rows % 64 == 0;
    // Deferred reductions (two-stream bf16 path, workspace permitting): the second stages of the block's column sums - c_fc bias, ln_2,
    // in_proj bias, ln_1 (+ c_proj bias of the last block) - are recorded while the block is enqueued and launched ONCE at its end; every
    // producer gets its own region of d->ws for its partials.  (Each of those small launches was a kernel the stream's next kernel waited
    // for while the other tower's persistent GEMM held the CUs: ~90 us apiece in the step, four per block.)
    static const bool defer_env = [] { const char* e = sc_debug_env("SC_BLOCK_DEFER_REDUCE"); return !(e && e[0] == '0'); }();
    const ReduceRegions rr = reduce_regions(rows, d->batch, W, MLP);
    const bool defer = two && grouped && defer_env && d->ws_bytes >= rr.total;   // grouped: no other reduction of this file is enqueued from here on the side stream
    ScReduceJobs jobs;
    struct DeferGuard {   // an early return (SC_TRY) must not leave the thread recording
        bool on;
        ~DeferGuard() { if (on) sc_reduce_defer_cancel(); }
    } guard{defer};
    if (defer) sc_reduce_defer_begin(&jobs);
    char* const ws0 = (char*)d->ws;
    void* const ws_fc1 = defer ? ws0 + rr.fc1 : d->ws;
    void* const ws_ln2 = defer ? ws0 + rr.ln2 : d->ws;
    void* const ws_attn = defer ? ws0 + rr.attn : d->ws;
    void* const ws_ln1 = defer ? ws0 + rr.ln1 : d->ws;
    const size_t ws_fc1_bytes = defer ? rr.ln2 - rr.fc1 : d->ws_bytes, ws_ln2_bytes = defer ? rr.attn - rr.ln2 : d->ws_bytes;
    const size_t ws_attn_bytes = defer ? rr.ln1 - rr.attn : d->ws_bytes, ws_ln1_bytes = defer ? rr.total - rr.ln1 : d->ws_bytes;

    // ---- MLP half: c_proj, GELU', c_fc
    if (!gate && !grouped) {
        SC_TRY(publish());
        SC_TRY(linear_dw(dt, rows, W, MLP, g, d->h_act, d->g_w_fc2, acc, wsw, wsw_bytes, ss));
    }
    if (!d->b_fc2_done) {   // the last block only; reads the fp32 dx_out: main stream.  Reduced at once (its partials would share a region with ln_1's)
        if (defer) sc_reduce_defer_cancel();
        SC_TRY(sc_colsum(dx_out, SC_F32, rows, W, W, d->g_b_fc2, acc, d->ws, d->ws_bytes, stream));
        if (defer) sc_reduce_defer_begin(&jobs);
    }
    static const bool fuse_cs = [] { const char* e = sc_debug_env("SC_BLOCK_FUSE_CS"); return !(e && e[0] == '0'); }();   // =0: separate column-sum passes (A/B runs)
    const bool fcs = bf && fuse_cs;
    EpiParams e = epi_plain();
    const bool ug = bf && fcs && unfuse_gelu();
    if (!ug) {
        e.dgelu_pre = d->h_pre; e.ld_aux = MLP;
        if (fcs) {   // c_fc's bias gradient = column sums of d_h: taken in the epilogue that produces d_h (main stream, main workspace)
            e.colsum = d->g_b_fc1; e.colsum_ws = ws_fc1; e.colsum_ws_bytes = ws_fc1_bytes; e.colsum_accumulate = acc;
        }
    }
    SC_TRY(linear_dx(dt, rows, W, MLP, g, d->w_fc2, d->wt_fc2, d->d_h, e, st, d->tile_tickets));                 // d_h = (dx_out W2) * gelu'(h_pre)
    if (ug) SC_TRY(sc_dgelu_mul_colsum_bf16(d->d_h, d->h_pre, rows, MLP, d->g_b_fc1, acc, d->ws, d->ws_bytes, st));
    if (!gate && !grouped) {
        SC_TRY(publish());
        SC_TRY(linear_dw(dt, rows, MLP, W, d->d_h, d->ln2_out, d->g_w_fc1, acc, wsw, wsw_bytes, ss));
        if (!fcs) SC_TRY(sc_colsum(d->d_h, dt, rows, MLP, MLP, d->g_b_fc1, acc, wsw, wsw_bytes, (void*)ss));
    }
    SC_TRY(linear_dx(dt, rows, MLP, W, d->d_h, d->w_fc1, d->wt_fc1, d->d_ln, epi_plain(), st, d->tile_tickets)); // d ln_2 output
    // dx_mid = dx_out + LN2'(d_ln); the same kernel emits the operand copy and the out_proj bias gradient (column sums of dx_mid).
    // With a side stream the copy must not land in the buffer `g` that the side stream may still be reading (internal-cast case).
    SC_REQUIRE(!(two && bf && g == d->d_res_t), SC_ERR_ARG, "sc_block_bwd_async: pass dx_out_t (the internal cast reuses d_res_t)");
    SC_TRY(sc_layernorm_bwd(d->d_ln, dt, d->x_mid, d->ln2_mean, d->ln2_rstd, d->ln2_g, rows, W, dx_out, d->dx_mid, bf ? d->d_res_t : nullptr,
                            d->g_ln2_g, d->g_ln2_b, d->g_b_o, acc, ws_ln2, ws_ln2_bytes, stream));
    const void* gm = bf ? (const void*)d->d_res_t : (const void*)d->dx_mid;
    // ---- attention half: out_proj, attention, in_proj
    if (!gate && !grouped) {
        SC_TRY(publish());
        SC_TRY(linear_dw(dt, rows, W, W, gm, d->attn_out, d->g_w_o, acc, wsw, wsw_bytes, ss));
    }
    SC_TRY(linear_dx(dt, rows, W, W, gm, d->w_o, d->wt_o, d->d_attn, epi_plain(), st, d->tile_tickets));
    if (gate && !grouped) {   // everything the three GEMMs read (g, h_act, d_h, ln2_out, d_res_t, attn_out) is final here
        SC_TRY(publish());
        SC_TRY(linear_dw(dt, rows, W, MLP, g, d->h_act, d->g_w_fc2, acc, wsw, wsw_bytes, ss));
        SC_TRY(linear_dw(dt, rows, MLP, W, d->d_h, d->ln2_out, d->g_w_fc1, acc, wsw, wsw_bytes, ss));
        if (!fcs) SC_TRY(sc_colsum(d->d_h, dt, rows, MLP, MLP, d->g_b_fc1, acc, wsw, wsw_bytes, (void*)ss));
        SC_TRY(linear_dw(dt, rows, W, W, gm, d->attn_out, d->g_w_o, acc, wsw, wsw_bytes, ss));
    }
    // in_proj's bias gradient = column sums of d_qkv: taken by the attention backward while dq / dk / dv are in registers
    if (dt == SC_F32 && !sc_attention_f32_bwd_fits_lds(d->seq))   // the one-workgroup-per-head fp32 backward holds S x S scores and dP in LDS
        SC_TRY(sc_attention_f32_composed_bwd((const float*)d->qkv, (const float*)d->d_attn, (float*)d->d_qkv, d->batch, d->seq, W, d->heads, d->causal, d->ws, d->ws_bytes, st));
    else SC_TRY(sc_attention_bwd_stats(d->qkv, d->attn_out, d->attn_lse, d->d_attn, d->d_qkv, dt, d->batch, d->seq, W, d->heads, d->causal, fcs ? d->g_b_qkv : nullptr, acc,
                                       ws_attn, ws_attn_bytes, stream));
    SC_TRY(publish());
    if (grouped) {   // dW of c_proj, c_fc, out_proj, in_proj: every operand (g, h_act, d_h, ln2_out, gm, attn_out, d_qkv, ln1_out) is final here
        const int64_t pm[4] = {W, MLP, W, 3 * W}, pn[4] = {MLP, W, W, W};
        const void* pa[4] = {g, d->d_h, gm, d->d_qkv};
        const void* pb[4] = {d->h_act, d->ln2_out, d->attn_out, d->ln1_out};
        float* pc[4] = {d->g_w_fc2, d->g_w_fc1, d->g_w_o, d->g_w_qkv};
        SC_TRY(sc_gemm_bf16_tn_group_launch(4, pm, pn, rows, pa, pm, pb, pn, pc, pn, 1.f, acc ? 1.f : 0.f, wsw, wsw_bytes, ss));
    } else {
        SC_TRY(linear_dw(dt, rows, 3 * W, W, d->d_qkv, d->ln1_out, d->g_w_qkv, acc, wsw, wsw_bytes, ss));
    }
    if (!fcs) SC_TRY(sc_colsum(d->d_qkv, dt, rows, 3 * W, 3 * W, d->g_b_qkv, acc, wsw, wsw_bytes, (void*)ss));
    SC_TRY(linear_dx(dt, rows, 3 * W, W, d->d_qkv, d->w_qkv, d->wt_qkv, d->d_ln, epi_plain(), st, d->tile_tickets));
    // dx_in = dx_mid + LN1'(d_ln)
    SC_TRY(sc_layernorm_bwd(d->d_ln, dt, d->x_in, d->ln1_mean, d->ln1_rstd, d->ln1_g, rows, W, d->dx_mid, dx_in, bf ? dx_in_t : nullptr, d->g_ln1_g,
                            d->g_ln1_b, d->g_below_b_fc2, acc, ws_ln1, ws_ln1_bytes, stream));
    if (defer) {
        guard.on = false;
        SC_TRY(sc_reduce_defer_flush(st));
    }
    return SC_OK;
}

}  // namespace

extern "C" int sc_event_create(void** event_out) {
    SC_REQUIRE(event_out != nullptr, SC_ERR_ARG, "sc_event_create: null output");
    hipEvent_t e = nullptr;
    const hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (rc != hipSuccess) return sc_set_error((int)rc, "sc_event_create: %s", hipGetErrorString(rc));
    *event_out = (void*)e;
    return SC_OK;
}

extern "C" int sc_event_destroy(void* event) {
    if (!event) return SC_OK;
    const hipError_t rc = hipEventDestroy((hipEvent_t)event);
    return rc == hipSuccess ? SC_OK : sc_set_error((int)rc, "sc_event_destroy: %s", hipGetErrorString(rc));
}

extern "C" int sc_block_bwd(const sc_block_desc* d, const float* dx_out, const void* dx_out_t, float* dx_in, void* dx_in_t, void* stream) {
    return block_bwd_impl(d, dx_out, dx_out_t, dx_in, dx_in_t, (hipStream_t)stream, nullptr);
}

extern "C" int sc_block_bwd_async(const sc_block_desc* d, const float* dx_out, const void* dx_out_t, float* dx_in, void* dx_in_t, void* stream,
                                  void* side_stream) {
    return block_bwd_impl(d, dx_out, dx_out_t, dx_in, dx_in_t, (hipStream_t)stream, (hipStream_t)side_stream);
}
