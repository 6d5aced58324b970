/*
 * sparsify_hip.h - C ABI of libsparsify_hip.so, the MI355X (gfx950) kernels behind the
 * CLIP-on-COCO training step of noostale/sparsify-clip.
 *
 * The reference has no FFI: its hot path reaches ATen/cuBLAS/cuDNN through PyTorch.  Each entry
 * point below therefore cites the reference line whose arithmetic it replaces
 * (/root/reference/sparsify_clip.py unless noted; encoder internals live in the un-vendored
 * open-clip-torch==2.29.0, reference environment.yml:191, call sites :768-769).
 *
 * Conventions
 *   - plain pointers and sizes only; every buffer (inputs, outputs, workspace) is owned by the caller
 *     and is DEVICE memory unless a parameter is documented as host;
 *   - row-major, leading dimension in ELEMENTS passed explicitly where a matrix is not dense;
 *   - every call enqueues work on `stream` (a hipStream_t passed as void*) and returns without
 *     synchronising; scalar results are written to device memory;
 *   - return 0 on success, negative SC_ERR_* for host-side argument errors, positive = hipError_t;
 *     sc_last_error() returns thread-local text; nothing is printed, nothing throws;
 *   - the library keeps no device memory, streams or events between calls: whatever outlives a call (workspaces, the events of
 *     sc_block_bwd_async) is created, passed in and destroyed by the caller, so calls are re-entrant across host threads;
 *   - reductions use fixed-order partials (no float atomics): results are bit-stable run to run.
 */
#ifndef SPARSIFY_HIP_H
#define SPARSIFY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SC_ABI_VERSION 6

enum { SC_OK = 0, SC_ERR_ARG = -1, SC_ERR_SHAPE = -2, SC_ERR_DTYPE = -3, SC_ERR_ALIGN = -4,
       SC_ERR_WORKSPACE = -5, SC_ERR_NO_DEVICE = -6 };
enum { SC_F32 = 0, SC_BF16 = 1 };

const char* sc_last_error(void);
int sc_abi_version(void);
/* size in bytes of the ABI structs, for binding self-checks: 0 = sc_block_desc, 1 = sc_gemm_epilogue */
size_t sc_abi_sizeof(int which);
int sc_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * GEMM  (replaces every torch `linear` / `mm` / `conv2d(stride=kernel)` on the path, SURVEY 2.3 K1,K4,K6-K8,K10,K12)
 * ---------------------------------------------------------------------------------------------- */
typedef struct sc_gemm_epilogue {
    /* out = act( alpha*acc + bias[n] ) * gelu'(dgelu_pre[m][n]) + resid[m][n] + beta*out_old ; each stage optional */
    float alpha, beta;
    const float* bias;        /* [N] fp32 or NULL */
    void* pre_out;            /* [M,N] activation dtype (fp32 in sc_gemm_f32, bf16 in sc_gemm_bf16_nt): value before the activation, or NULL */
    int32_t act;              /* 0 none, 1 exact erf GELU (nn.GELU) */
    int32_t resid_dtype;      /* dtype of resid (SC_F32 or SC_BF16) */
    const void* resid;        /* [M,N] added after the activation, or NULL */
    const void* dgelu_pre;    /* [M,N] activation dtype: multiply by GELU'(this), or NULL (backward of K7) */
    int64_t ld_aux;           /* leading dimension of pre_out / resid / dgelu_pre */
    /* optional: colsum[n] = (colsum_accumulate ? colsum[n] : 0) + sum_m out[m][n] of the values as stored (the bias gradient of
     * the layer that produced the GEMM's A operand when `out` is a pre-activation gradient, e.g. d_h = (dY W2) * GELU'(h)).
     * Fused into the epilogue where the kernel supports it, otherwise a pass over `out`; fixed-order partial sums in colsum_ws
     * (>= 4 * N * ceil(M / 128) bytes, or >= 4096 * N bytes for the fallback pass). */
    float* colsum;            /* [N] fp32 or NULL */
    void* colsum_ws;
    uint64_t colsum_ws_bytes;
    int32_t colsum_accumulate;
    int32_t reserved_;
    /* optional, sc_gemm_bf16_nt only: 64 bytes of device memory, 64-byte aligned, ZERO when first used, owned by the caller and
     * used by the GEMM calls of ONE stream at a time.  With it the persistent kernel's workgroups claim their 256x256 tiles from
     * per-XCD ticket counters instead of walking fixed lists, so a launch that shares the GPU with other kernels (another stream,
     * an RCCL collective) finishes when the work is done rather than when the most delayed workgroup has walked its list.  The
     * kernel leaves the words zero again when it ends; the result does not depend on which workgroup computed which tile. */
    void* tile_tickets;
} sc_gemm_epilogue;

/* fp32 MFMA GEMM, any operand orientation: C[M,N] = op(A) * op(B).
 * trans_a == 0: A is [M,K] (lda >= K); trans_a == 1: A is [K,M] (lda >= M).
 * trans_b == 0: B is [K,N] (ldb >= N); trans_b == 1: B is [N,K] (ldb >= K)  (torch linear weight layout). */
int sc_gemm_f32(int trans_a, int trans_b, int64_t m, int64_t n, int64_t k,
                const float* a, int64_t lda, const float* b, int64_t ldb, float* c, int64_t ldc,
                const sc_gemm_epilogue* epi /* NULL = plain product */, void* stream);

/* bf16 MFMA GEMM, fp32 accumulate.  "NT": C[M,N] = A[M,K] * B[N,K]^T, both K-contiguous
 * (activation x torch-layout weight).  K % 64 == 0, lda/ldb % 8 == 0, 16-byte aligned bases.
 * out_dtype selects a bf16 or fp32 C. */
int sc_gemm_bf16_nt(int64_t m, int64_t n, int64_t k, const void* a, int64_t lda, const void* b, int64_t ldb,
                    void* c, int64_t ldc, int out_dtype, const sc_gemm_epilogue* epi, void* stream);
/* "TN": C[M,N] (fp32) = alpha * sum_r A[r][m] * B[r][n] + beta*C, A is [R,M], B is [R,N] (weight gradients:
 * dW = dY^T X).  M % 8 == 0, N % 8 == 0.  The contraction is split over workgroups when M*N is small; the
 * fp32 partial slabs go to ws (sc_gemm_bf16_tn_workspace_bytes) and are summed in a fixed order. */
size_t sc_gemm_bf16_tn_workspace_bytes(int64_t m, int64_t n, int64_t r);
int sc_gemm_bf16_tn(int64_t m, int64_t n, int64_t r, const void* a, int64_t lda, const void* b, int64_t ldb,
                    float* c, int64_t ldc, float alpha, float beta, void* ws, size_t ws_bytes, void* stream);
/* Same, plus colsum_a[m] = colsum_beta * colsum_a[m] + sum_r A[r][m] from the same staged tiles: with A = dY this is the
 * bias gradient of the linear layer whose weight gradient the call computes (autograd of nn.Linear under reference
 * sparsify_clip.py:965 loss.backward()).  colsum_a: M fp32, 16-byte aligned. */
int sc_gemm_bf16_tn_colsum(int64_t m, int64_t n, int64_t r, const void* a, int64_t lda, const void* b, int64_t ldb,
                           float* c, int64_t ldc, float alpha, float beta, float* colsum_a, float colsum_beta,
                           void* ws, size_t ws_bytes, void* stream);

/* Up to four TN problems with ONE contraction length r (r % 64 == 0) in one launch: C_k = alpha * A_k^T B_k + beta * C_k.  The four
 * weight gradients of a transformer block (same rows, different dY / X: autograd of the four nn.Linear under reference
 * sparsify_clip.py:965) share the chip, so the contraction is split 2-5 ways (one set of partial slabs in ws, one fixed-order reduce)
 * instead of 7-28 ways per problem.  All pointer arrays are HOST arrays of nprob entries. */
size_t sc_gemm_bf16_tn_group_workspace_bytes(int nprob, const int64_t* m, const int64_t* n, int64_t r);
int sc_gemm_bf16_tn_group(int nprob, const int64_t* m, const int64_t* n, int64_t r, const void* const* a, const int64_t* lda,
                          const void* const* b, const int64_t* ldb, float* const* c, const int64_t* ldc, float alpha, float beta,
                          void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Loss head on [B,E] fp32 embeddings  (sparsify_clip.py:110-187, :334-355, :772-773, :804)
 * All *_fwd_bwd calls write the scalar loss to loss_out[0] (device) and the gradient of
 * (grad_scale * loss) to the d_* buffers (overwriting).  d_* may be NULL to skip the backward.
 * ---------------------------------------------------------------------------------------------- */
size_t sc_loss_workspace_bytes(int64_t b, int64_t e);

/* contrastive_loss(image_embeds, text_embeds, temperature)  :110-132.  d_temp (device float, may be NULL)
 * receives d loss / d temperature for the learnable-temperature configuration (:716-717). */
int sc_contrastive_fwd_bwd(const float* img, const float* txt, int64_t b, int64_t e, float temperature,
                           float grad_scale, float* loss_out, float* d_img, float* d_txt, float* d_temp,
                           void* ws, size_t ws_bytes, void* stream);
/* lunif_loss(x, t)  :159-164  (log-mean-exp over all i<j of -t*||xi-xj||^2, via Gram + fused exp). */
int sc_lunif_fwd_bwd(const float* x, int64_t b, int64_t e, float t, float grad_scale, float* loss_out,
                     float* d_x, void* ws, size_t ws_bytes, void* stream);
/* lalign_loss(x, y, alpha)  :186-187 */
int sc_lalign_fwd_bwd(const float* x, const float* y, int64_t b, int64_t e, float alpha, float grad_scale,
                      float* loss_out, float* d_x, float* d_y, void* ws, size_t ws_bytes, void* stream);
/* sparsify_loss(x)  :166-176 */
int sc_sparsify_fwd_bwd(const float* x, int64_t b, int64_t e, float grad_scale, float* loss_out, float* d_x,
                        void* ws, size_t ws_bytes, void* stream);
/* Row-block ("sharded") forms of the two O(B^2) terms for data-parallel training: the caller holds the gathered [b,e] embeddings and
 * owns rows [row0, row0 + bm) of them.  *_stats runs before the ranks exchange their statistics, *_grad after it; every rank
 * then has the same loss value and its own rows of the gradients - (b / bm) times less work per rank than the replicated call.
 * Needs b % 64 == 0, e % 128 == 0 (e <= 1024), row0 % 64 == 0, bm % 64 == 0 (SC_ERR_SHAPE otherwise: use the replicated call);
 * ws as for the replicated calls (sc_loss_workspace_bytes(b, e)); nothing is kept between the two calls.
 *   contrastive (:110-132): stats[3*bm] = row LSE of my image rows | column LSE of my text columns | diagonal logits of my rows;
 *     _grad takes the three statistics gathered over all ranks ([b] each, rank-major = row order), writes the loss (identical on
 *     every rank), my rows of d_img / d_txt and, if d_temp_part != NULL, my rows' part of d loss / d temperature (SUM over ranks).
 *   lunif (:159-164): _stats gives my rows of the row sums of W and of W X plus s_part = their sum; _grad takes every rank's s_part. */
int sc_contrastive_rows_stats(const float* img, const float* txt, int64_t b, int64_t e, int64_t row0, int64_t bm, float temperature,
                              float* stats, void* ws, size_t ws_bytes, void* stream);
int sc_contrastive_rows_grad(const float* img, const float* txt, int64_t b, int64_t e, int64_t row0, int64_t bm, float temperature,
                             float grad_scale, const float* row_lse, const float* col_lse, const float* diag, float* loss_out,
                             float* d_img_rows, float* d_txt_rows, float* d_temp_part, void* ws, size_t ws_bytes, void* stream);
int sc_lunif_rows_stats(const float* x, int64_t b, int64_t e, int64_t row0, int64_t bm, float t, float* rowsum_rows, float* wx_rows,
                        float* s_part, void* ws, size_t ws_bytes, void* stream);
int sc_lunif_rows_grad(const float* x_rows, int64_t b, int64_t bm, int64_t e, float t, float grad_scale, const float* s_parts, int64_t nparts,
                       const float* rowsum_rows, const float* wx_rows, float* loss_out, float* dx_rows, void* stream);
/* Row L2 normalisation y = x / max(||x||, eps); eps = 0 reproduces :772-773 (no clamp), eps = 1e-12 F.normalize (:804).
 * inv_norm [B] is saved for the backward. */
int sc_l2norm_fwd(const float* x, int64_t b, int64_t e, float eps, float* y, float* inv_norm, void* stream);
int sc_l2norm_bwd(const float* y, const float* inv_norm, const float* dy, int64_t b, int64_t e, float* dx, void* stream);
/* compute_centroids_only + F.normalize  (:334-355, :804): c = normalize((a+b)/2). */
int sc_centroid_fwd(const float* a, const float* b_, int64_t b, int64_t e, float* c, float* inv_norm, void* stream);
/* backward of the above: given dc, ACCUMULATES into d_a and d_b (d += ...). */
int sc_centroid_bwd(const float* c, const float* inv_norm, const float* dc, int64_t b, int64_t e,
                    float* d_a, float* d_b, void* stream);
/* out[i] += alpha * in[i]  (fp32), n elements; used to combine loss-term gradients. */
int sc_axpy_f32(int64_t n, float alpha, const float* x, float* y, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Eval: retrieval ranks on device (sparsify_clip.py:357-416) without sort/.tolist().
 * score is [N,N] fp32 (rows = text queries, :628).  rank_fwd[i] = #{j : s[i][j] > s[i][i]} (+ ties with j < i),
 * rank_bwd[j] likewise down column j.  top1_fwd/top1_bwd = argmax (first occurrence), int32.
 * ---------------------------------------------------------------------------------------------- */
int sc_retrieval_ranks(const float* score, int64_t n, int32_t* rank_fwd, int32_t* rank_bwd,
                       int32_t* top1_fwd, int32_t* top1_bwd, void* stream);

/* Eval geometry + recall counts in one call, no [N,N] matrix and no host round trip per metric (sparsify_clip.py:418-436 compute_gap,
 * :438-457 mean angular value, :508-528 mean cosine of true pairs, :382-392 / :404-414 R@1/5/10):
 *   out10[0] = | mean_i img_i - mean_i txt_i |                out10[3] = mean_i <img_i, txt_i>
 *   out10[1] = mean_{i != j} <img_i, img_j>   out10[2] = same for txt   (via |sum_i x_i|^2 - sum_i |x_i|^2)
 *   out10[4..6] / out10[7..9] = #queries with rank < 1 / 5 / 10 in rank_fwd / rank_bwd (either may be NULL -> zeros) */
size_t sc_eval_metrics_workspace_bytes(int64_t n, int64_t e);
int sc_eval_metrics(const float* img, const float* txt, int64_t n, int64_t e, const int32_t* rank_fwd, const int32_t* rank_bwd,
                    float* out10, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Encoder building blocks (open_clip ViT / text transformer)
 * `dtype` is the activation/GEMM-operand dtype (SC_BF16 or SC_F32); the residual stream, LayerNorm
 * statistics, biases, LayerNorm affine parameters and all parameter gradients are fp32.
 * ---------------------------------------------------------------------------------------------- */
/* LayerNorm(eps 1e-5, affine) over rows of a fp32 [rows,width] matrix -> y (dtype).  K3 */
int sc_layernorm_fwd(const float* x, int64_t rows, int64_t width, const float* gamma, const float* beta,
                     void* y, int dtype, float* mean, float* rstd, void* stream);
/* dx = (dres ? dres : 0) + LN'(dy); dgamma/dbeta accumulate (+=) when accumulate != 0. ws >= 3*1024*width floats */
int sc_layernorm_bwd(const void* dy, int dtype, const float* x, const float* mean, const float* rstd,
                     const float* gamma, int64_t rows, int64_t width, const float* dres, float* dx,
                     void* dx_cast /* optional copy of dx in `dtype`, the next GEMM's operand */,
                     float* dgamma, float* dbeta, float* dx_colsum /* optional [width]: (+)= column sums of dx (a bias gradient) */,
                     int accumulate, void* ws, size_t ws_bytes, void* stream);
/* softmax(QK^T/sqrt(64) [+causal mask]) V for packed qkv [B*S, 3*W] (nn.MultiheadAttention in_proj layout). K5 */
int sc_attention_fwd(const void* qkv, void* out, int dtype, int64_t batch, int64_t seq, int64_t width, int64_t heads,
                     int causal, void* stream);
int sc_attention_bwd(const void* qkv, const void* d_out, void* d_qkv, int dtype, int64_t batch, int64_t seq,
                     int64_t width, int64_t heads, int causal, void* stream);
/* Same, plus colsum[n] (+)= column sums of d_qkv over all rows: the in_proj bias gradient (autograd of nn.MultiheadAttention's
 * in_proj under reference sparsify_clip.py:965).  The bf16 MFMA kernels accumulate them while they hold dq / dk / dv in
 * registers ([batch][3*width] fp32 partials in ws, >= 4*batch*3*width bytes, then a fixed-order reduce); other paths run a
 * pass over d_qkv (ws >= 4096*3*width*4 bytes). */
int sc_attention_bwd_colsum(const void* qkv, const void* d_out, void* d_qkv, int dtype, int64_t batch, int64_t seq, int64_t width,
                            int64_t heads, int causal, float* colsum, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* The same pair in the flash-attention form: the forward also leaves the softmax statistics of every query row,
 *   lse[(b * heads + h) * seq + i] = -(m_i log2 e + log2 l_i)   (fp32 [batch * heads * seq];  P_ij = 2^(s_ij / 8 * log2 e + lse_i)),
 * and the backward takes them together with the forward's output `out` (delta_i = dO_i . O_i), so it needs neither the row maxima nor
 * the row sums again: on long sequences (bf16, 80 < seq <= 272: one workgroup per head, scores recomputed per key tile) pass 1 of the
 * backward is one sweep over the keys instead of three.  sc_attention_uses_stats(dtype, seq) != 0 says whether the forward writes
 * `lse` for a shape (otherwise it is left untouched and the backward ignores it: both calls then behave as the plain pair above);
 * colsum may be NULL (no bias gradient; ws unused then).  Replaces nn.MultiheadAttention's core behind reference
 * sparsify_clip.py:768-769 for ViT-L/14's 257-token image sequences (BASELINE config 5). */
int sc_attention_uses_stats(int dtype, int64_t seq);
int sc_attention_fwd_stats(const void* qkv, void* out, float* lse, int dtype, int64_t batch, int64_t seq, int64_t width, int64_t heads,
                           int causal, void* stream);
int sc_attention_bwd_stats(const void* qkv, const void* out, const float* lse, const void* d_out, void* d_qkv, int dtype, int64_t batch,
                           int64_t seq, int64_t width, int64_t heads, int causal, float* colsum, int accumulate, void* ws, size_t ws_bytes,
                           void* stream);
/* column sums: out[n] (+)= sum_r x[r][n]  (bias gradients).  ws >= 1024*n floats */
int sc_colsum(const void* x, int dtype, int64_t rows, int64_t n, int64_t ld, float* out, int accumulate,
              void* ws, size_t ws_bytes, void* stream);
/* K1: images fp32 [B,3,R,R] -> patches [B*g*g, kpad] (dtype), g = R/P, column (c,ky,kx); columns >= 3*P*P zero */
int sc_im2col(const float* images, int64_t batch, int64_t res, int64_t patch, int64_t kpad, void* out, int dtype, void* stream);
/* K2: x[b,0,:] = cls + pos[0]; x[b,1+p,:] = patch_out[b*g*g+p,:] + pos[1+p]  -> fp32 [B*S, W] */
int sc_vit_tokens_fwd(const void* patch_out, int dtype, const float* cls, const float* pos, int64_t batch, int64_t seq,
                      int64_t width, float* x, void* stream);
/* backward: d_patch_out (dtype), d_cls [W] and d_pos [S,W] (+= when accumulate) from dx fp32 [B*S,W] */
int sc_vit_tokens_bwd(const float* dx, int64_t batch, int64_t seq, int64_t width, void* d_patch_out, int dtype,
                      float* d_cls, float* d_pos, int accumulate, void* stream);
/* K9: x[b,s,:] = tok_emb[tokens[b,s]] + pos[s] -> fp32 [B*S,W]; tokens int64 */
int sc_text_embed_fwd(const int64_t* tokens, const float* tok_emb, const float* pos, int64_t batch, int64_t seq,
                      int64_t width, int64_t vocab, float* x, void* stream);
/* backward: d_pos [S,W] (+=) and the token-embedding scatter-add.  `order` [n_sorted] lists flat positions b*S+s sorted by
 * token id (stable), `sorted_tokens` = tokens[order]; each d_tok_emb [vocab,W] row receives the sum of its run in that fixed
 * order (no atomics).  Positions whose dx row is known to be zero (after EOT under the causal mask) may be left out. */
int sc_text_embed_bwd(const float* dx, const int64_t* sorted_tokens, const int64_t* order, int64_t n_sorted,
                      int64_t batch, int64_t seq, int64_t width, int64_t vocab, float* d_tok_emb, float* d_pos,
                      int accumulate, void* stream);
/* Index bookkeeping of sc_text_embed_bwd on the device: sorted_keys / order [batch * seq] = the flat positions b * seq + s sorted (stable)
 * by token id, with key `vocab` for the positions behind a caption's EOT (s > eot[b]: exactly-zero gradient under the causal mask of
 * open_clip's text tower, ignored by sc_text_embed_bwd).  tokens [batch, seq] contiguous.  ws: sc_token_sort_workspace_bytes(batch * seq, vocab). */
size_t sc_token_sort_workspace_bytes(int64_t n, int64_t vocab);
int sc_token_sort(const int64_t* tokens, const int32_t* eot, int64_t batch, int64_t seq, int64_t vocab, int64_t* sorted_keys, int64_t* order,
                  void* ws, size_t ws_bytes, void* stream);
/* eot[b] = argmax_s tokens[b,s] (first occurrence)  (text_global_pool 'argmax') */
int sc_argmax_tokens(const int64_t* tokens, int64_t batch, int64_t seq, int32_t* eot, void* stream);
/* K10 gather: out[b,:] = x[b*seq + (idx ? idx[b] : 0), :]  (fp32) ; scatter is its adjoint into a ZEROED dx */
int sc_pool_gather(const float* x, const int32_t* idx, int64_t batch, int64_t seq, int64_t width, float* out, void* stream);
int sc_pool_scatter(const float* d_out, const int32_t* idx, int64_t batch, int64_t seq, int64_t width, float* dx, void* stream);
/* dst(bf16)[i] = src(fp32)[i] */
int sc_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
/* dst(bf16) [cols,rows] = transpose(src fp32 [rows,cols]) */
int sc_transpose_cast_bf16(const float* src, int64_t rows, int64_t cols, void* dst, void* stream);

/* The same for a table of matrices in ONE launch (the ~100 [in,out] weight copies rebuilt after every optimiser step):
 * table = DEVICE int64 [n_items][5] = {src pointer, dst pointer, rows, cols, first_block}, first_block = running sum of
 * ceil(rows/32) * ceil(cols/32) over the preceding items (ascending), total_blocks = that sum over all items. */
int sc_transpose_cast_bf16_batch(const int64_t* table, int64_t n_items, int64_t total_blocks, void* stream);

/* One pre-LN residual attention block, forward and backward (K3-K8).  All pointers device. */
typedef struct sc_block_desc {
    int64_t batch, seq, width, heads, mlp_width;
    int32_t dtype;      /* SC_BF16 or SC_F32 */
    int32_t causal;
    /* parameters, fp32 master (LN affine, biases always read from here) */
    const float *ln1_g, *ln1_b, *b_qkv, *b_o, *ln2_g, *ln2_b, *b_fc1, *b_fc2;
    /* GEMM weights in `dtype`, torch layout [out,in]; for SC_F32 these are the fp32 masters themselves */
    const void *w_qkv, *w_o, *w_fc1, *w_fc2;
    /* transposed [in,out] copies in `dtype` (SC_BF16 only, used by the backward); NULL for SC_F32 */
    const void *wt_qkv, *wt_o, *wt_fc1, *wt_fc2;
    /* saved activations (written by fwd, read by bwd) */
    const float* x_in;  /* [rows,W] fp32 residual stream entering the block */
    float* x_mid;       /* [rows,W] fp32 after attention residual */
    float* x_out;       /* [rows,W] fp32 block output */
    void *ln1_out, *qkv, *attn_out, *ln2_out, *h_pre, *h_act;   /* dtype */
    float *ln1_mean, *ln1_rstd, *ln2_mean, *ln2_rstd;           /* [rows] */
    /* backward: gradients of parameters, fp32, accumulated (+=) when accumulate != 0 */
    float *g_ln1_g, *g_ln1_b, *g_w_qkv, *g_b_qkv, *g_w_o, *g_b_o, *g_ln2_g, *g_ln2_b, *g_w_fc1, *g_b_fc1, *g_w_fc2, *g_b_fc2;
    int32_t accumulate;
    int32_t b_fc2_done;  /* != 0: g_b_fc2 was already produced by the block above (see g_below_b_fc2) */
    /* backward scratch (dtype unless noted): d_h [rows,mlp], d_ln [rows,W], d_qkv [rows,3W], d_attn [rows,W],
     * d_res_t [rows,W] (GEMM-operand copy of the fp32 residual gradients; unused for SC_F32) */
    void *d_h, *d_ln, *d_qkv, *d_attn, *d_res_t;
    float* dx_mid;      /* [rows,W] fp32 scratch */
    void* ws;           /* reduction workspace */
    size_t ws_bytes;
    /* optional: the c_proj bias gradient of the block BELOW (= column sums of dx_in), fused into this block's ln_1 backward */
    float* g_below_b_fc2;
    /* workspace of the weight-gradient side stream of sc_block_bwd_async (same size rule as ws); may be NULL otherwise */
    void* ws_side;
    size_t ws_side_bytes;
    /* sc_block_bwd_async only: four caller-owned HIP events (sc_event_create) that order `stream` and `side_stream` inside the
     * call.  The library keeps none of its own: descriptors used from different host threads or devices get different events;
     * descriptors enqueued one after the other from ONE thread may share a set. */
    void* events[4];
    /* optional: tile tickets for the NT GEMMs the call enqueues on `stream` (sc_gemm_epilogue.tile_tickets: 64 zeroed, 64-byte
     * aligned bytes); descriptors whose calls are enqueued on one stream may share them.  NULL = fixed tile lists. */
    void* tile_tickets;
    /* optional (ABI 5): fp32 [batch * heads * seq] softmax statistics of the block's attention, written by sc_block_fwd and read by
     * sc_block_bwd when sc_attention_uses_stats(dtype, seq) (sc_attention_fwd_stats / _bwd_stats); NULL = the plain attention pair */
    float* attn_lse;
} sc_block_desc;

size_t sc_block_workspace_bytes(int64_t rows, int64_t width, int64_t mlp_width, int dtype);
/* HIP events (hipEventDisableTiming) on the CURRENT device for sc_block_desc.events; the caller destroys them. */
int sc_event_create(void** event_out /* host */);
int sc_event_destroy(void* event);
/* x_out = block(x_in)  (open_clip ResidualAttentionBlock.forward) */
int sc_block_fwd(const sc_block_desc* d, void* stream);
/* dx_in (fp32 [rows,W]) = (d block / d x_in)^T dx_out; parameter grads into g_*.  dx_out_t / dx_in_t are optional copies
 * of dx_out / dx_in in `dtype` (SC_BF16 only): pass the previous call's dx_in_t as the next call's dx_out_t to skip a cast. */
int sc_block_bwd(const sc_block_desc* d, const float* dx_out, const void* dx_out_t, float* dx_in, void* dx_in_t, void* stream);
/* Same, with the weight-gradient GEMMs (dW = dY^T X) and the bias column sums of d_h / d_qkv enqueued on `side_stream`: they are
 * off the critical path of the activation gradients, so they overlap the HBM-bound kernels of `stream` (LayerNorm, attention).
 * Ordering inside the call is by the caller-owned events d->events[0..3]; the CALLER must (a) make `stream` wait for the side work of an
 * earlier call before a later call reuses the same scratch buffers (d_h, d_qkv, d_res_t, dx_out_t/dx_in_t, ws_side) - i.e. give
 * consecutive blocks alternating scratch sets - and (b) join `side_stream` before reading the parameter gradients. */
int sc_block_bwd_async(const sc_block_desc* d, const float* dx_out, const void* dx_out_t, float* dx_in, void* dx_in_t, void* stream,
                       void* side_stream);

/* ------------------------------------------------------------------------------------------------
 * Input stage on the device: the reference's image transforms (sparsify_clip.py:1003-1016) from raw uint8 RGB pixels to the
 * normalised fp32 [N,3,S,S] batch:  train = RandomResizedCrop((S,S)) -> RandomHorizontalFlip -> ToTensor -> Normalize(mean,std),
 * test = Resize((S,S)) -> ToTensor -> Normalize.  "Resize" is Pillow's antialiased two-pass BILINEAR resampler (what torchvision
 * runs on the PIL images of CocoCaptions), restated bit for bit: 22-bit fixed-point taps, 8-bit intermediate image.
 *   src        packed uint8 images, HWC interleaved RGB; image n starts at byte offset[n]
 *   dims       int32 [N][6] = {H, W, top, left, h, w}: full size and the crop box (the whole image for Resize)
 *   flip       int32 [N] (may be NULL): != 0 mirrors the resized image horizontally
 *   tmp        uint8 workspace; image n's intermediate (h x S x 3 bytes) lives at tmp_offset[n]; max_box_h = max_n h
 * All arrays are DEVICE memory; the random crop boxes / flips are drawn on the host (reference :1009-1010 via torchvision). */
int sc_image_resample_normalize(const uint8_t* src, const int64_t* offset, const int32_t* dims, const int32_t* flip,
                                const int64_t* tmp_offset, int64_t n, int64_t max_box_h, int64_t out_size,
                                float mean_r, float mean_g, float mean_b, float std_r, float std_g, float std_b,
                                void* tmp, float* out, void* stream);

/* HOST function (no GPU call): item k = rows[k] rows of row_bytes[k] bytes, src_row_stride[k] bytes apart, starting at src[k] (a crop
 * box inside a decoded image) -> dst + dst_offset[k], contiguous; `threads` host threads share the items.  Fills the pinned staging
 * buffer of sc_image_resample_normalize's H2D copy (the reference's DataLoader workers + default collate, sparsify_clip.py:1060-1065). */
int sc_host_gather_rows(int64_t n, const void* const* src, const int64_t* src_row_stride, const int64_t* rows, const int64_t* row_bytes,
                        void* dst, const int64_t* dst_offset, int64_t dst_bytes, int threads);

/* ------------------------------------------------------------------------------------------------
 * Optimiser: torch.optim.AdamW defaults over one flat fp32 parameter buffer (sparsify_clip.py:730, :962/966).
 * p *= 1 - lr*wd; m,v moments; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps).  `step` is the 1-based step count.
 * shadow_bf16 (may be NULL) receives a bf16 copy of the updated parameters.
 * grad_scale multiplies g before use (1.0 normally).
 * ---------------------------------------------------------------------------------------------- */
int sc_adamw_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int64_t step, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ModifiedResNet ("RN50", the `model:` of every reference YAML; open_clip ModifiedResNet / Bottleneck / AttentionPool2d behind
 * sparsify_clip.py:685-689, :768).  Activations are NHWC = row-major [B*H*W, C] matrices in the compute dtype (SC_BF16 / SC_F32):
 * a 1x1 convolution is sc_gemm_*_nt as it stands, a 3x3 convolution is sc_im2col3x3 + the same GEMM, its input gradient the GEMM
 * against the transposed weight + sc_col2im3x3, its weight gradient sc_gemm_*_tn.
 * ---------------------------------------------------------------------------------------------- */
/* out[(b,yo,xo)][tap*C + c] = x[b][yo*stride+ky-1][xo*stride+kx-1][c] (tap = 3ky+kx, zero padding, columns >= 9C zero; kpad >= 9C).
 * in_nchw_f32 != 0: x is the fp32 image tensor [B,C,H,W] (stem), else an NHWC activation of `dtype`.  out is `dtype`. */
int sc_im2col3x3(const void* x, int in_nchw_f32, int dtype, int64_t batch, int64_t h, int64_t w, int64_t c, int64_t stride, int64_t kpad,
                 void* out, void* stream);
int sc_col2im3x3(const void* dcols, int dtype, int64_t batch, int64_t h, int64_t w, int64_t c, int64_t stride, int64_t kpad, void* dx, void* stream);
/* The same convolution (stride 1) WITHOUT a patch matrix: an implicit GEMM over an activation with a one-pixel zero border,
 * a_halo [batch][h+2][w+2][cin] bf16 (written by sc_bn_apply / sc_bn_bwd_apply with their halo option); b [n][9*cin] tap-major,
 * cin = 64 * 2^j; c [batch*h*w][n] (out_dtype).  With b = the weight re-arranged [cin][(8 - tap)*cout + co] and a_halo = the bordered
 * output gradient the same call gives the input gradient.  epi: bias / residual only (NULL = none). */
int sc_conv3x3_bf16(const void* a_halo, const void* b, void* c, int out_dtype, int64_t batch, int64_t h, int64_t w, int64_t cin, int64_t n,
                    const sc_gemm_epilogue* epi, void* stream);
/* Weight gradient of sc_conv3x3_bf16, one launch, no patch matrix: dw[cout][tap * cin + c] (tap-major, fp32) = alpha * sum over the bordered
 * rows r of dz_halo[r][:]^T x_halo[r + shift(tap)][:] + beta * dw.  dz_halo [batch][h+2][w+2][cout] has a ZERO border; x_halo
 * [batch][h+2][w+2][cin] additionally has w + 3 rows of zeros readable before and after the image (the shifted reads of the first and last
 * rows).  ws: sc_gemm_bf16_tn_workspace_bytes(cout, 9 * cin, batch * (h+2) * (w+2)) bytes. */
int sc_conv3x3_dw_bf16(const void* dz_halo, const void* x_halo, float* dw, int64_t batch, int64_t h, int64_t w, int64_t cout, int64_t cin,
                       float alpha, float beta, void* ws, size_t ws_bytes, void* stream);
/* nn.AvgPool2d(k) (k = stride; h, w multiples of k) and its backward */
int sc_avgpool_fwd(const void* x, int dtype, int64_t batch, int64_t h, int64_t w, int64_t c, int64_t k, void* y, void* stream);
int sc_avgpool_bwd(const void* dy, int dtype, int64_t batch, int64_t h, int64_t w, int64_t c, int64_t k, void* dx, void* stream);
/* nn.BatchNorm2d in training mode over x [rows, c].  sc_bn_stats: this rank's shifted sums stats[3c] (sum(x-k), sum (x-k)^2, k);
 * sc_bn_finish: from nparts such triples of rows_per_part rows each (1 part without data parallelism; the gathered triples of all
 * ranks for a synchronised BatchNorm) the batch mean and 1/sqrt(biased var + eps), and the running statistics update of torch
 * (momentum, unbiased variance; NULL skips it).  sc_bn_apply: y = act((x-mean)*rstd*gamma + beta (+ res)), act = ReLU if relu.
 * Backward: sc_bn_bwd_stats gives sums[2c] = sum g | sum g*xhat over this rank's rows (g = dy masked by y > 0 if relu; SUM them over
 * ranks for a synchronised BatchNorm); sc_bn_bwd_apply writes dx, optionally dres = g (the gradient of the residual input),
 * and dgamma / dbeta (+= if accumulate) from `sums` with total_rows = rows of the whole batch.  y == NULL with relu (forward WITHOUT a
 * residual, channel count a multiple of 4 that the vector kernels take): the ReLU mask is recomputed from x, gamma, beta - one tensor
 * less to read in both backward passes.  halo_h / halo_w != 0 (sc_bn_apply: y; sc_bn_bwd_apply: dx): the output is written into a
 * bordered image [batch][halo_h+2][halo_w+2][halo_c] with halo_c >= c channels (interior pixels and the first c channels only; the
 * caller zeroes the rest: a 32-channel activation inside a 64-channel image meets sc_conv3x3_bf16's cin = 64 * 2^j) - the operand
 * layout of sc_conv3x3_bf16.  halo_h == halo_w == 0: compact rows, halo_c ignored. */
size_t sc_bn_workspace_bytes(int64_t rows, int64_t c);
int sc_bn_stats(const void* x, int dtype, int64_t rows, int64_t c, float* stats, void* ws, size_t ws_bytes, void* stream);
int sc_bn_finish(const float* stats, int64_t nparts, int64_t c, int64_t rows_per_part, float eps, float momentum, float* mean, float* rstd,
                 float* running_mean, float* running_var, void* stream);
int sc_bn_apply(const void* x, int dtype, int64_t rows, int64_t c, const float* mean, const float* rstd, const float* gamma, const float* beta,
                const void* res, int relu, int64_t halo_h, int64_t halo_w, int64_t halo_c, void* y, void* stream);
int sc_bn_bwd_stats(const void* dy, const void* y, const void* x, int dtype, int64_t rows, int64_t c, const float* mean, const float* rstd,
                    const float* gamma, const float* beta, int relu, float* sums, void* ws, size_t ws_bytes, void* stream);
int sc_bn_bwd_apply(const void* dy, const void* y, const void* x, int dtype, int64_t rows, int64_t c, const float* mean, const float* rstd,
                    const float* gamma, const float* beta, const float* sums, int64_t total_rows, int relu, int accumulate,
                    int64_t halo_h, int64_t halo_w, int64_t halo_c, void* dx, void* dres, float* dgamma, float* dbeta, void* stream);
/* AttentionPool2d token assembly: tokens[b][0] = mean_p x[b][p] + pos[0], tokens[b][p+1] = x[b][p] + pos[p+1]; backward w.r.t. x */
int sc_attnpool_tokens_fwd(const void* x, int dtype, const float* pos, int64_t batch, int64_t hw, int64_t c, void* tokens, void* stream);
int sc_attnpool_tokens_bwd(const void* dtokens, int dtype, int64_t batch, int64_t hw, int64_t c, void* dx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPARSIFY_HIP_H */
