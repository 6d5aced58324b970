"""Thin tensor-level wrappers over the C ABI (one function per entry point of include/sparsify_hip.h).

Every wrapper takes CUDA(ROCm) tensors, enqueues on the current stream and returns tensors; nothing here
computes on the host and nothing falls back to torch when the library is missing.
"""
from __future__ import annotations

import ctypes
import os

import torch

from ._lib import LIB, SC_BF16, SC_F32, BlockDesc, GemmEpilogue, ScError, ptr, require_gpu, sc_dtype, stream_ptr

_ws_cache = {}


def _workspace(nbytes: int, device, tag: str = "ws") -> torch.Tensor:
    """Grow-only byte workspace per (device, tag, stream); reused across calls on the same stream."""
    key = (str(device), tag, torch.cuda.current_stream().cuda_stream)   # one workspace per stream: the towers run concurrently
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def make_epilogue(alpha=1.0, beta=0.0, bias=None, pre_out=None, act=0, resid=None, dgelu_pre=None, ld_aux=0, colsum=None,
                  colsum_accumulate=False, rows=None, tile_tickets=None):
    """colsum ([N] fp32): also receive the column sums of the stored output; `rows` (the GEMM's M) sizes the partial-sum workspace.
    tile_tickets (>= 16 zeroed int32, one stream at a time): dynamic tile order of the persistent NT kernel."""
    e = GemmEpilogue()
    e.alpha, e.beta, e.act = float(alpha), float(beta), int(act)
    e.bias = bias.data_ptr() if bias is not None else None
    e.pre_out = pre_out.data_ptr() if pre_out is not None else None
    e.resid = resid.data_ptr() if resid is not None else None
    e.resid_dtype = sc_dtype(resid.dtype) if resid is not None else SC_F32
    e.dgelu_pre = dgelu_pre.data_ptr() if dgelu_pre is not None else None
    e.ld_aux = int(ld_aux)
    ws = None
    if colsum is not None:
        if rows is None:
            raise ScError("make_epilogue: colsum needs rows (the GEMM's M)")
        n = colsum.numel()
        ws = torch.empty(max(4096 * n, ((rows + 127) // 128) * n) * 4, dtype=torch.uint8, device=colsum.device)
        e.colsum, e.colsum_ws, e.colsum_ws_bytes, e.colsum_accumulate = colsum.data_ptr(), ws.data_ptr(), ws.numel(), int(bool(colsum_accumulate))
    e.tile_tickets = tile_tickets.data_ptr() if tile_tickets is not None else None
    e._keepalive = (bias, pre_out, resid, dgelu_pre, colsum, ws, tile_tickets)   # the struct only holds raw pointers
    return e


# ------------------------------------------------------------------------------------------------ GEMM
def gemm_f32(a, b, trans_a=False, trans_b=False, out=None, epi: GemmEpilogue | None = None):
    require_gpu(a, "a", torch.float32), require_gpu(b, "b", torch.float32)
    m, k = (a.shape[1], a.shape[0]) if trans_a else a.shape
    n = b.shape[0] if trans_b else b.shape[1]
    kb = b.shape[1] if trans_b else b.shape[0]
    if kb != k:
        raise ScError(f"gemm_f32: inner dimensions differ ({k} vs {kb})")
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=a.device)
    LIB.call("sc_gemm_f32", int(trans_a), int(trans_b), m, n, k, ptr(a), a.shape[1], ptr(b), b.shape[1], ptr(out), out.shape[1],
             ctypes.byref(epi) if epi is not None else None, stream_ptr())
    return out


def gemm_bf16_nt(a, b, out_dtype=torch.bfloat16, out=None, epi: GemmEpilogue | None = None):
    require_gpu(a, "a", torch.bfloat16), require_gpu(b, "b", torch.bfloat16)
    m, k = a.shape
    n = b.shape[0]
    if b.shape[1] != k:      # the C ABI sees pointers and sizes only: a wrong operand here is an out-of-bounds read on the card
        raise ScError(f"gemm_bf16_nt: a is [{m}, {k}] but b is {list(b.shape)} (expected [n, {k}])")
    if out is None:
        out = torch.empty(m, n, dtype=out_dtype, device=a.device)
    elif tuple(out.shape) != (m, n):
        raise ScError(f"gemm_bf16_nt: out is {list(out.shape)}, expected [{m}, {n}]")
    LIB.call("sc_gemm_bf16_nt", m, n, k, ptr(a), k, ptr(b), b.shape[1], ptr(out), n, sc_dtype(out.dtype),
             ctypes.byref(epi) if epi is not None else None, stream_ptr())
    return out


def gemm_bf16_tn(a, b, out=None, alpha=1.0, beta=0.0, colsum_out=None, colsum_beta=0.0):
    """out[M,N] = alpha * a[R,M]^T b[R,N] + beta*out (fp32); with colsum_out ([M] fp32) also
    colsum_out = colsum_beta * colsum_out + a.sum(0) from the same kernel (the bias gradient when a = dY)."""
    require_gpu(a, "a", torch.bfloat16), require_gpu(b, "b", torch.bfloat16)
    r, m = a.shape
    n = b.shape[1]
    if b.shape[0] != r:
        raise ScError(f"gemm_bf16_tn: a has {r} rows, b has {b.shape[0]}")
    if out is None:
        out = torch.zeros(m, n, dtype=torch.float32, device=a.device)
    elif tuple(out.shape) != (m, n):
        raise ScError(f"gemm_bf16_tn: out is {list(out.shape)}, expected [{m}, {n}]")
    nbytes = LIB.raw("sc_gemm_bf16_tn_workspace_bytes")(m, n, r)
    ws = _workspace(nbytes, a.device)
    if colsum_out is None:
        LIB.call("sc_gemm_bf16_tn", m, n, r, ptr(a), m, ptr(b), n, ptr(out), n, float(alpha), float(beta), ptr(ws), ws.numel(), stream_ptr())
        return out
    require_gpu(colsum_out, "colsum_out", torch.float32)
    if colsum_out.numel() != m:
        raise ScError(f"gemm_bf16_tn: colsum_out has {colsum_out.numel()} elements, expected {m}")
    LIB.call("sc_gemm_bf16_tn_colsum", m, n, r, ptr(a), m, ptr(b), n, ptr(out), n, float(alpha), float(beta), ptr(colsum_out), float(colsum_beta),
             ptr(ws), ws.numel(), stream_ptr())
    return out


def gemm_bf16_tn_group(problems, alpha=1.0, beta=0.0):
    """problems: up to four (a [R,M] bf16, b [R,N] bf16, out [M,N] fp32) sharing R: out = alpha * a^T b + beta * out, ONE launch."""
    n = len(problems)
    r = problems[0][0].shape[0]
    I64, VP = ctypes.c_int64 * n, ctypes.c_void_p * n
    m = I64(*[p[0].shape[1] for p in problems])
    nn = I64(*[p[1].shape[1] for p in problems])
    for a, b, c in problems:
        require_gpu(a, "a", torch.bfloat16), require_gpu(b, "b", torch.bfloat16), require_gpu(c, "out", torch.float32)
        if a.shape[0] != r or b.shape[0] != r or tuple(c.shape) != (a.shape[1], b.shape[1]):
            raise ScError("gemm_bf16_tn_group: shapes disagree")
    nbytes = LIB.raw("sc_gemm_bf16_tn_group_workspace_bytes")(n, m, nn, r)
    ws = _workspace(nbytes, problems[0][0].device)
    LIB.call("sc_gemm_bf16_tn_group", n, m, nn, r, VP(*[p[0].data_ptr() for p in problems]), m, VP(*[p[1].data_ptr() for p in problems]), nn,
             VP(*[p[2].data_ptr() for p in problems]), nn, float(alpha), float(beta), ptr(ws), ws.numel(), stream_ptr())
    return [p[2] for p in problems]


# ------------------------------------------------------------------------------------------------ loss head
def _loss_ws(b, e, device):
    n = LIB.raw("sc_loss_workspace_bytes")(b, e)
    return _workspace(n, device, "loss")


def contrastive_fwd_bwd(img, txt, temperature, grad_scale=1.0, need_grad=True, need_dtemp=False):
    require_gpu(img, "img", torch.float32), require_gpu(txt, "txt", torch.float32)
    b, e = img.shape
    ws = _loss_ws(b, e, img.device)
    loss = torch.empty(1, dtype=torch.float32, device=img.device)
    d_img = torch.empty_like(img) if need_grad else None
    d_txt = torch.empty_like(txt) if need_grad else None
    d_temp = torch.empty(1, dtype=torch.float32, device=img.device) if (need_grad and need_dtemp) else None
    LIB.call("sc_contrastive_fwd_bwd", ptr(img), ptr(txt), b, e, float(temperature), float(grad_scale), ptr(loss), ptr(d_img), ptr(d_txt),
             ptr(d_temp), ptr(ws), ws.numel(), stream_ptr())
    return loss, d_img, d_txt, d_temp


def lunif_fwd_bwd(x, t=2.0, grad_scale=1.0, need_grad=True):
    require_gpu(x, "x", torch.float32)
    b, e = x.shape
    ws = _loss_ws(b, e, x.device)
    loss = torch.empty(1, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x) if need_grad else None
    LIB.call("sc_lunif_fwd_bwd", ptr(x), b, e, float(t), float(grad_scale), ptr(loss), ptr(dx), ptr(ws), ws.numel(), stream_ptr())
    return loss, dx


def lalign_fwd_bwd(x, y, alpha=2.0, grad_scale=1.0, need_grad=True):
    require_gpu(x, "x", torch.float32), require_gpu(y, "y", torch.float32)
    b, e = x.shape
    ws = _loss_ws(b, e, x.device)
    loss = torch.empty(1, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x) if need_grad else None
    dy = torch.empty_like(y) if need_grad else None
    LIB.call("sc_lalign_fwd_bwd", ptr(x), ptr(y), b, e, float(alpha), float(grad_scale), ptr(loss), ptr(dx), ptr(dy), ptr(ws), ws.numel(),
             stream_ptr())
    return loss, dx, dy


def sparsify_fwd_bwd(x, grad_scale=1.0, need_grad=True):
    require_gpu(x, "x", torch.float32)
    b, e = x.shape
    ws = _loss_ws(b, e, x.device)
    loss = torch.empty(1, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x) if need_grad else None
    LIB.call("sc_sparsify_fwd_bwd", ptr(x), b, e, float(grad_scale), ptr(loss), ptr(dx), ptr(ws), ws.numel(), stream_ptr())
    return loss, dx

# ---- row-block ("sharded") forms: the rank owns rows [row0, row0 + bm) of the gathered batch (include/sparsify_hip.h)
def loss_rows_supported(b, e, row0, bm) -> bool:
    """Shapes the row-block loss head takes (otherwise the caller evaluates the replicated loss head)."""
    return b % 64 == 0 and b >= 128 and e % 128 == 0 and 128 <= e <= 1024 and row0 % 64 == 0 and bm % 64 == 0 and bm >= 64 and row0 + bm <= b


def contrastive_rows_stats(img, txt, row0, bm, temperature):
    """-> stats [3, bm]: row LSE of my image rows, column LSE of my text columns, diagonal logits of my rows."""
    require_gpu(img, "img", torch.float32), require_gpu(txt, "txt", torch.float32)
    b, e = img.shape
    ws = _loss_ws(b, e, img.device)
    stats = torch.empty(3, bm, dtype=torch.float32, device=img.device)
    LIB.call("sc_contrastive_rows_stats", ptr(img), ptr(txt), b, e, int(row0), int(bm), float(temperature), ptr(stats), ptr(ws), ws.numel(), stream_ptr())
    return stats


def contrastive_rows_grad(img, txt, row0, bm, temperature, row_lse, col_lse, diag, grad_scale=1.0, need_dtemp=False):
    """row_lse / col_lse / diag: [B] gathered over the ranks.  -> loss [1], my rows of d_img / d_txt, my part of d_temp (or None)."""
    b, e = img.shape
    ws = _loss_ws(b, e, img.device)
    loss = torch.empty(1, dtype=torch.float32, device=img.device)
    d_img = torch.empty(bm, e, dtype=torch.float32, device=img.device)
    d_txt = torch.empty(bm, e, dtype=torch.float32, device=img.device)
    d_temp = torch.empty(1, dtype=torch.float32, device=img.device) if need_dtemp else None
    LIB.call("sc_contrastive_rows_grad", ptr(img), ptr(txt), b, e, int(row0), int(bm), float(temperature), float(grad_scale), ptr(row_lse), ptr(col_lse),
             ptr(diag), ptr(loss), ptr(d_img), ptr(d_txt), ptr(d_temp), ptr(ws), ws.numel(), stream_ptr())
    return loss, d_img, d_txt, d_temp


def lunif_rows_stats(x, row0, bm, t=2.0):
    """-> my rows of the row sums of W [bm] and of W X [bm, e], and their sum s_part [1]."""
    require_gpu(x, "x", torch.float32)
    b, e = x.shape
    ws = _loss_ws(b, e, x.device)
    rowsum = torch.empty(bm, dtype=torch.float32, device=x.device)
    wx = torch.empty(bm, e, dtype=torch.float32, device=x.device)
    s_part = torch.empty(1, dtype=torch.float32, device=x.device)
    LIB.call("sc_lunif_rows_stats", ptr(x), b, e, int(row0), int(bm), float(t), ptr(rowsum), ptr(wx), ptr(s_part), ptr(ws), ws.numel(), stream_ptr())
    return rowsum, wx, s_part


def lunif_rows_grad(x_rows, b, t, grad_scale, s_parts, rowsum, wx):
    """s_parts: every rank's s_part [world].  -> loss [1] (identical on every rank), my rows of the gradient."""
    bm, e = x_rows.shape
    loss = torch.empty(1, dtype=torch.float32, device=x_rows.device)
    dx = torch.empty_like(x_rows)
    LIB.call("sc_lunif_rows_grad", ptr(x_rows), int(b), bm, e, float(t), float(grad_scale), ptr(s_parts), s_parts.numel(), ptr(rowsum), ptr(wx), ptr(loss),
             ptr(dx), stream_ptr())
    return loss, dx



def l2norm_fwd(x, eps=0.0, out=None):
    require_gpu(x, "x", torch.float32)
    b, e = x.shape
    y = torch.empty_like(x) if out is None else require_gpu(out, "out", torch.float32)
    inv = torch.empty(b, dtype=torch.float32, device=x.device)
    LIB.call("sc_l2norm_fwd", ptr(x), b, e, float(eps), ptr(y), ptr(inv), stream_ptr())
    return y, inv


def l2norm_bwd(y, inv, dy):
    require_gpu(dy, "dy", torch.float32)
    dx = torch.empty_like(y)
    LIB.call("sc_l2norm_bwd", ptr(y), ptr(inv), ptr(dy), y.shape[0], y.shape[1], ptr(dx), stream_ptr())
    return dx


def centroid_fwd(a, b):
    require_gpu(a, "a", torch.float32), require_gpu(b, "b", torch.float32)
    c = torch.empty_like(a)
    inv = torch.empty(a.shape[0], dtype=torch.float32, device=a.device)
    LIB.call("sc_centroid_fwd", ptr(a), ptr(b), a.shape[0], a.shape[1], ptr(c), ptr(inv), stream_ptr())
    return c, inv


def centroid_bwd_accumulate(c, inv, dc, d_a, d_b):
    LIB.call("sc_centroid_bwd", ptr(c), ptr(inv), ptr(dc), c.shape[0], c.shape[1], ptr(d_a), ptr(d_b), stream_ptr())


def axpy_(y, alpha, x):
    LIB.call("sc_axpy_f32", y.numel(), float(alpha), ptr(x), ptr(y), stream_ptr())
    return y


def retrieval_ranks(score):
    require_gpu(score, "score", torch.float32)
    n = score.shape[0]
    outs = [torch.empty(n, dtype=torch.int32, device=score.device) for _ in range(4)]
    LIB.call("sc_retrieval_ranks", ptr(score), n, *[ptr(o) for o in outs], stream_ptr())
    return outs  # rank_fwd, rank_bwd, top1_fwd, top1_bwd


def eval_metrics(img, txt, rank_fwd=None, rank_bwd=None):
    """-> device fp32 [10]: gap, mean angular (img, txt), mean true-pair cosine, R@1/5/10 hit counts forward / backward."""
    require_gpu(img, "img", torch.float32), require_gpu(txt, "txt", torch.float32)
    n, e = img.shape
    out = torch.empty(10, dtype=torch.float32, device=img.device)
    ws = _workspace(LIB.raw("sc_eval_metrics_workspace_bytes")(n, e), img.device, "eval")
    LIB.call("sc_eval_metrics", ptr(img), ptr(txt), n, e, ptr(rank_fwd), ptr(rank_bwd), ptr(out), ptr(ws), ws.numel(), stream_ptr())
    return out


# ------------------------------------------------------------------------------------------------ encoder pieces
def layernorm_fwd(x, gamma, beta, out_dtype, out=None):
    require_gpu(x, "x", torch.float32)
    rows, w = x.shape
    y = out if out is not None else torch.empty(rows, w, dtype=out_dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    LIB.call("sc_layernorm_fwd", ptr(x), rows, w, ptr(gamma), ptr(beta), ptr(y), sc_dtype(out_dtype), ptr(mean), ptr(rstd), stream_ptr())
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, dres=None, want_cast=False, dgamma=None, dbeta=None, accumulate=False, dx_colsum=None):
    rows, w = x.shape
    dx = torch.empty_like(x)
    dx_cast = torch.empty(rows, w, dtype=dy.dtype, device=x.device) if want_cast else None
    if dgamma is None:
        dgamma = torch.zeros(w, dtype=torch.float32, device=x.device)
        dbeta = torch.zeros(w, dtype=torch.float32, device=x.device)
    ws = _workspace(1024 * 3 * w * 4, x.device)
    LIB.call("sc_layernorm_bwd", ptr(dy), sc_dtype(dy.dtype), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), rows, w, ptr(dres), ptr(dx), ptr(dx_cast),
             ptr(dgamma), ptr(dbeta), ptr(dx_colsum), int(accumulate), ptr(ws), ws.numel(), stream_ptr())
    return dx, dx_cast, dgamma, dbeta


def attention_uses_stats(dtype, seq) -> bool:
    """Whether the forward writes softmax statistics for this shape (sc_attention_fwd_stats / _bwd_stats: bf16, 80 < seq <= 272)."""
    return bool(LIB.raw("sc_attention_uses_stats")(sc_dtype(dtype), int(seq)))


def attention_fwd(qkv, batch, seq, heads, causal, lse=None):
    """out; with `lse` (fp32 [batch * heads * seq]) the flash-attention form: the statistics are written where attention_uses_stats says so."""
    rows, w3 = qkv.shape
    w = w3 // 3
    out = torch.empty(rows, w, dtype=qkv.dtype, device=qkv.device)
    if lse is not None:
        LIB.call("sc_attention_fwd_stats", ptr(qkv), ptr(out), ptr(lse), sc_dtype(qkv.dtype), batch, seq, w, heads, int(causal), stream_ptr())
    else:
        LIB.call("sc_attention_fwd", ptr(qkv), ptr(out), sc_dtype(qkv.dtype), batch, seq, w, heads, int(causal), stream_ptr())
    return out


def attention_bwd(qkv, d_out, batch, seq, heads, causal, colsum_out=None, colsum_accumulate=False, out=None, lse=None):
    """d_qkv; with colsum_out ([3*width] fp32) also the column sums of d_qkv over all rows (the in_proj bias gradient); with the forward's
    `out` and `lse` the flash-attention form of the backward (sc_attention_bwd_stats)."""
    w = qkv.shape[1] // 3
    d_qkv = torch.empty_like(qkv)
    if out is not None and lse is not None:
        ws = _workspace(max(batch, 4096) * 3 * w * 4, qkv.device) if colsum_out is not None else None
        LIB.call("sc_attention_bwd_stats", ptr(qkv), ptr(out), ptr(lse), ptr(d_out), ptr(d_qkv), sc_dtype(qkv.dtype), batch, seq, w, heads, int(causal),
                 ptr(colsum_out), int(bool(colsum_accumulate)), ptr(ws), ws.numel() if ws is not None else 0, stream_ptr())
        return d_qkv
    if colsum_out is None:
        LIB.call("sc_attention_bwd", ptr(qkv), ptr(d_out), ptr(d_qkv), sc_dtype(qkv.dtype), batch, seq, w, heads, int(causal), stream_ptr())
        return d_qkv
    ws = _workspace(max(batch, 4096) * 3 * w * 4, qkv.device)
    LIB.call("sc_attention_bwd_colsum", ptr(qkv), ptr(d_out), ptr(d_qkv), sc_dtype(qkv.dtype), batch, seq, w, heads, int(causal), ptr(colsum_out),
             int(bool(colsum_accumulate)), ptr(ws), ws.numel(), stream_ptr())
    return d_qkv


def colsum(x, out=None, accumulate=False):
    rows, n = x.shape
    if out is None:
        out = torch.zeros(n, dtype=torch.float32, device=x.device)
    ws = _workspace(1024 * n * 4, x.device)
    LIB.call("sc_colsum", ptr(x), sc_dtype(x.dtype), rows, n, n, ptr(out), int(accumulate), ptr(ws), ws.numel(), stream_ptr())
    return out


def im2col(images, patch, kpad, out_dtype):
    require_gpu(images, "images", torch.float32)
    b, _, res, _ = images.shape
    g = res // patch
    out = torch.empty(b * g * g, kpad, dtype=out_dtype, device=images.device)
    LIB.call("sc_im2col", ptr(images), b, res, patch, kpad, ptr(out), sc_dtype(out_dtype), stream_ptr())
    return out


def vit_tokens_fwd(patch_out, cls, pos, batch, seq):
    w = patch_out.shape[1]
    x = torch.empty(batch * seq, w, dtype=torch.float32, device=patch_out.device)
    LIB.call("sc_vit_tokens_fwd", ptr(patch_out), sc_dtype(patch_out.dtype), ptr(cls), ptr(pos), batch, seq, w, ptr(x), stream_ptr())
    return x


def vit_tokens_bwd(dx, batch, seq, out_dtype, d_cls, d_pos, accumulate):
    w = dx.shape[1]
    d_patch = torch.empty(batch * (seq - 1), w, dtype=out_dtype, device=dx.device)
    LIB.call("sc_vit_tokens_bwd", ptr(dx), batch, seq, w, ptr(d_patch), sc_dtype(out_dtype), ptr(d_cls), ptr(d_pos), int(accumulate), stream_ptr())
    return d_patch


def text_embed_fwd(tokens, tok_emb, pos, out=None):
    require_gpu(tokens, "tokens", torch.int64)
    b, s = tokens.shape
    vocab, w = tok_emb.shape
    x = out if out is not None else torch.empty(b * s, w, dtype=torch.float32, device=tokens.device)
    LIB.call("sc_text_embed_fwd", ptr(tokens), ptr(tok_emb), ptr(pos), b, s, w, vocab, ptr(x), stream_ptr())
    return x


def text_embed_bwd(dx, sorted_tokens, order, batch, seq, d_tok_emb, d_pos, accumulate):
    vocab, w = d_tok_emb.shape
    LIB.call("sc_text_embed_bwd", ptr(dx), ptr(sorted_tokens), ptr(order), order.numel(), batch, seq, w, vocab, ptr(d_tok_emb), ptr(d_pos),
             int(accumulate), stream_ptr())


def token_sort(tokens, eot, vocab):
    """(sorted_keys, order) of sc_text_embed_bwd for tokens [B, S]: flat positions sorted stably by token id, positions behind the EOT keyed `vocab`."""
    b, s = tokens.shape
    n = b * s
    keys = torch.empty(n, dtype=torch.int64, device=tokens.device)
    order = torch.empty(n, dtype=torch.int64, device=tokens.device)
    ws = _workspace(LIB.raw("sc_token_sort_workspace_bytes")(n, vocab), tokens.device, "token_sort")
    LIB.call("sc_token_sort", ptr(tokens), ptr(eot), b, s, vocab, ptr(keys), ptr(order), ptr(ws), ws.numel(), stream_ptr())
    return keys, order


def argmax_tokens(tokens):
    b, s = tokens.shape
    eot = torch.empty(b, dtype=torch.int32, device=tokens.device)
    LIB.call("sc_argmax_tokens", ptr(tokens), b, s, ptr(eot), stream_ptr())
    return eot


def pool_gather(x, idx, batch, seq):
    w = x.shape[1]
    out = torch.empty(batch, w, dtype=torch.float32, device=x.device)
    LIB.call("sc_pool_gather", ptr(x), ptr(idx), batch, seq, w, ptr(out), stream_ptr())
    return out


def pool_scatter(d_out, idx, batch, seq, dx):
    """dx must be zeroed by the caller."""
    LIB.call("sc_pool_scatter", ptr(d_out), ptr(idx), batch, seq, d_out.shape[1], ptr(dx), stream_ptr())
    return dx


def image_resample_normalize(pixels, offset, dims, flip, tmp_offset, n, max_box_h, tmp_bytes, size, mean, std, out=None):
    """uint8 crop boxes -> normalised fp32 [n,3,size,size] (csrc/augment.hip); every array argument is a device tensor."""
    require_gpu(pixels, "pixels", torch.uint8)
    if out is None:
        out = torch.empty(n, 3, size, size, dtype=torch.float32, device=pixels.device)
    tmp = _workspace(int(tmp_bytes), pixels.device, "aug")
    LIB.call("sc_image_resample_normalize", ptr(pixels), ptr(offset), ptr(dims), ptr(flip), ptr(tmp_offset), int(n), int(max_box_h), int(size),
             float(mean[0]), float(mean[1]), float(mean[2]), float(std[0]), float(std[1]), float(std[2]), ptr(tmp), ptr(out), stream_ptr())
    return out


def cast_bf16(src, dst=None):
    require_gpu(src, "src", torch.float32)
    if dst is None:
        dst = torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
    LIB.call("sc_cast_f32_to_bf16", ptr(src), ptr(dst), src.numel(), stream_ptr())
    return dst


def transpose_cast_bf16(src, dst=None):
    rows, cols = src.shape
    if dst is None:
        dst = torch.empty(cols, rows, dtype=torch.bfloat16, device=src.device)
    LIB.call("sc_transpose_cast_bf16", ptr(src), rows, cols, ptr(dst), stream_ptr())
    return dst


def transpose_cast_bf16_batch(pairs):
    """pairs: list of (src fp32 [r,c], dst bf16 [c,r]) -> a launcher that redoes all the transposes in ONE kernel launch."""
    rows, first = [], 0
    for src, dst in pairs:
        r, c = src.shape
        rows.append([src.data_ptr(), dst.data_ptr(), r, c, first])
        first += ((r + 31) // 32) * ((c + 31) // 32)
    table = torch.tensor(rows, dtype=torch.int64, device=pairs[0][0].device)

    def run():
        LIB.call("sc_transpose_cast_bf16_batch", ptr(table), len(rows), first, stream_ptr())
    run.table = table
    return run


def adamw_step(p, g, m, v, shadow, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    LIB.call("sc_adamw_step", ptr(p), ptr(g), ptr(m), ptr(v), ptr(shadow), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
             float(weight_decay), int(step), float(grad_scale), stream_ptr())


def block_workspace_bytes(rows, width, mlp_width, dtype) -> int:
    return LIB.raw("sc_block_workspace_bytes")(rows, width, mlp_width, sc_dtype(dtype))


def block_fwd(desc: BlockDesc):
    LIB.call("sc_block_fwd", ctypes.byref(desc), stream_ptr())


def block_bwd(desc: BlockDesc, dx_out, dx_out_t, dx_in, dx_in_t, side_stream=None):
    if side_stream is None:
        LIB.call("sc_block_bwd", ctypes.byref(desc), ptr(dx_out), ptr(dx_out_t), ptr(dx_in), ptr(dx_in_t), stream_ptr())
    else:
        LIB.call("sc_block_bwd_async", ctypes.byref(desc), ptr(dx_out), ptr(dx_out_t), ptr(dx_in), ptr(dx_in_t), stream_ptr(),
                 ctypes.c_void_p(side_stream.cuda_stream))


# ------------------------------------------------------------------------------------------------ ModifiedResNet pieces (csrc/conv.hip)
def im2col3x3(x, batch, h, w, c, stride, kpad, out_dtype, nchw_images=False):
    """NHWC activation [B*H*W, C] (or the fp32 image tensor [B,C,H,W] when nchw_images) -> [B*Ho*Wo, kpad] patch matrix."""
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    if x.numel() != batch * h * w * c or kpad < 9 * c:
        raise ScError(f"im2col3x3: x has {x.numel()} elements, expected {batch} x {h} x {w} x {c} (kpad {kpad} >= {9 * c})")
    out = torch.empty(batch * ho * wo, kpad, dtype=out_dtype, device=x.device)
    LIB.call("sc_im2col3x3", ptr(x), int(nchw_images), sc_dtype(out_dtype), batch, h, w, c, stride, kpad, ptr(out), stream_ptr())
    return out


def col2im3x3(dcols, batch, h, w, c, stride, kpad):
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    if tuple(dcols.shape) != (batch * ho * wo, kpad) or kpad < 9 * c:
        raise ScError(f"col2im3x3: dcols is {list(dcols.shape)}, expected [{batch * ho * wo}, {kpad}] with kpad >= {9 * c}")
    dx = torch.empty(batch * h * w, c, dtype=dcols.dtype, device=dcols.device)
    LIB.call("sc_col2im3x3", ptr(dcols), sc_dtype(dcols.dtype), batch, h, w, c, stride, kpad, ptr(dx), stream_ptr())
    return dx


def avgpool_fwd(x, batch, h, w, c, k):
    if x.numel() != batch * h * w * c:
        raise ScError(f"avgpool_fwd: x has {x.numel()} elements, expected {batch} x {h} x {w} x {c}")
    y = torch.empty(batch * (h // k) * (w // k), c, dtype=x.dtype, device=x.device)
    LIB.call("sc_avgpool_fwd", ptr(x), sc_dtype(x.dtype), batch, h, w, c, k, ptr(y), stream_ptr())
    return y


def avgpool_bwd(dy, batch, h, w, c, k):
    if dy.numel() != batch * (h // k) * (w // k) * c:
        raise ScError(f"avgpool_bwd: dy has {dy.numel()} elements, expected {batch} x {h // k} x {w // k} x {c}")
    dx = torch.empty(batch * h * w, c, dtype=dy.dtype, device=dy.device)
    LIB.call("sc_avgpool_bwd", ptr(dy), sc_dtype(dy.dtype), batch, h, w, c, k, ptr(dx), stream_ptr())
    return dx


def _bn_ws(rows, c, device):
    return _workspace(LIB.raw("sc_bn_workspace_bytes")(rows, c), device, "bn")


def bn_stats(x):
    """x [rows, C] -> this rank's statistics triple [3C] (shifted sums + shift)."""
    rows, c = x.shape
    stats = torch.empty(3 * c, dtype=torch.float32, device=x.device)
    ws = _bn_ws(rows, c, x.device)
    LIB.call("sc_bn_stats", ptr(x), sc_dtype(x.dtype), rows, c, ptr(stats), ptr(ws), ws.numel(), stream_ptr())
    return stats


def bn_finish(stats, nparts, c, rows_per_part, running_mean=None, running_var=None, eps=1e-5, momentum=0.1):
    mean = torch.empty(c, dtype=torch.float32, device=stats.device)
    rstd = torch.empty_like(mean)
    LIB.call("sc_bn_finish", ptr(stats), nparts, c, rows_per_part, float(eps), float(momentum), ptr(mean), ptr(rstd), ptr(running_mean), ptr(running_var),
             stream_ptr())
    return mean, rstd


def halo_buffer(batch, h, w, c, dtype, device):
    """Zeroed bordered NHWC image [batch, h+2, w+2, c] inside a flat buffer with (w + 3) slack rows at both ends (the shifted views of the
    per-tap weight-gradient GEMMs stay inside it).  -> (flat buffer, the image view)."""
    slack, rows = w + 3, batch * (h + 2) * (w + 2)
    flat = torch.zeros(rows + 2 * slack, c, dtype=dtype, device=device)
    return flat, flat[slack:slack + rows]


def _check_halo(halo, rows, c, what):
    if halo is None:
        return
    img, h, w = halo
    cs = img.shape[-1]      # the image's channel count: >= c (a 32-channel activation inside a 64-channel image; the rest stays zero)
    if rows % (h * w) or cs < c or img.numel() != (rows // (h * w)) * (h + 2) * (w + 2) * cs or img.dtype != torch.bfloat16 and img.dtype != torch.float32:
        raise ScError(f"{what}: the bordered image has {img.numel()} elements, expected [{rows // (h * w)}, {h + 2}, {w + 2}, >= {c}]")


def bn_apply(x, mean, rstd, gamma, beta, relu, res=None, halo=None):
    """halo = (image view of halo_buffer, h, w): write y into the bordered image instead of a compact [rows, c] tensor."""
    rows, c = x.shape
    y = torch.empty_like(x) if halo is None else halo[0]
    hh, hw, hc = (0, 0, 0) if halo is None else (halo[1], halo[2], halo[0].shape[-1])
    _check_halo(halo, rows, c, "bn_apply")
    LIB.call("sc_bn_apply", ptr(x), sc_dtype(x.dtype), rows, c, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(res), int(relu), hh, hw, hc, ptr(y), stream_ptr())
    return y


def bn_mask_from_x(c) -> bool:
    """Channel counts for which the BatchNorm backward can recompute the ReLU mask from x (y = None) instead of reading the stored output."""
    if c % 4:
        return False
    chunk = c if c <= 1024 else 1024
    lpr = chunk // 4
    return c % chunk == 0 and lpr & (lpr - 1) == 0


def bn_bwd_stats(dy, y, x, mean, rstd, relu, gamma=None, beta=None):
    rows, c = x.shape
    sums = torch.empty(2 * c, dtype=torch.float32, device=x.device)
    ws = _bn_ws(rows, c, x.device)
    LIB.call("sc_bn_bwd_stats", ptr(dy), ptr(y), ptr(x), sc_dtype(x.dtype), rows, c, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), int(relu), ptr(sums),
             ptr(ws), ws.numel(), stream_ptr())
    return sums


def bn_bwd_apply(dy, y, x, mean, rstd, gamma, sums, total_rows, relu, dgamma, dbeta, accumulate, want_dres=False, beta=None, halo=None):
    rows, c = x.shape
    dx = torch.empty_like(x) if halo is None else halo[0]
    hh, hw, hc = (0, 0, 0) if halo is None else (halo[1], halo[2], halo[0].shape[-1])
    _check_halo(halo, rows, c, "bn_bwd_apply")
    dres = torch.empty_like(x) if want_dres else None
    LIB.call("sc_bn_bwd_apply", ptr(dy), ptr(y), ptr(x), sc_dtype(x.dtype), rows, c, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(sums), int(total_rows),
             int(relu), int(accumulate), hh, hw, hc, ptr(dx), ptr(dres), ptr(dgamma), ptr(dbeta), stream_ptr())
    return dx, dres


def attnpool_tokens_fwd(x, pos, batch, hw):
    c = x.shape[1]
    t = torch.empty(batch * (hw + 1), c, dtype=x.dtype, device=x.device)
    LIB.call("sc_attnpool_tokens_fwd", ptr(x), sc_dtype(x.dtype), ptr(pos), batch, hw, c, ptr(t), stream_ptr())
    return t


def attnpool_tokens_bwd(dt, batch, hw):
    c = dt.shape[1]
    dx = torch.empty(batch * hw, c, dtype=dt.dtype, device=dt.device)
    LIB.call("sc_attnpool_tokens_bwd", ptr(dt), sc_dtype(dt.dtype), batch, hw, c, ptr(dx), stream_ptr())
    return dx


def conv3x3_bf16(a_halo, w_taps, batch, h, w, out_dtype=torch.bfloat16, epi: GemmEpilogue | None = None):
    """Implicit-GEMM 3x3 convolution (stride 1, padding 1): a_halo [batch, h+2, w+2, cin] bf16 with a zero border, w_taps [n, 9*cin]
    tap-major -> [batch*h*w, n].  No patch matrix."""
    require_gpu(a_halo, "a_halo", torch.bfloat16), require_gpu(w_taps, "w_taps", torch.bfloat16)
    cin = a_halo.shape[-1]
    n = w_taps.shape[0]
    if a_halo.numel() != batch * (h + 2) * (w + 2) * cin or w_taps.shape[1] != 9 * cin:
        raise ScError(f"conv3x3_bf16: a_halo has {a_halo.numel()} elements for [{batch}, {h + 2}, {w + 2}, {cin}], w_taps is {list(w_taps.shape)}")
    out = torch.empty(batch * h * w, n, dtype=out_dtype, device=a_halo.device)
    LIB.call("sc_conv3x3_bf16", ptr(a_halo), ptr(w_taps), ptr(out), sc_dtype(out_dtype), batch, h, w, cin, n,
             ctypes.byref(epi) if epi is not None else None, stream_ptr())
    return out


def conv3x3_dw_bf16(dz_img, x_flat, batch, h, w, out=None, beta=0.0):
    """Weight gradient of conv3x3_bf16 in one launch: dz_img = the image view of a halo_buffer holding dz (zero border), x_flat = the FLAT
    halo_buffer of the convolution's input (its zero slack rows are read by the shifted taps) -> [cout, 9 * cin] fp32, tap-major."""
    require_gpu(dz_img, "dz_img", torch.bfloat16), require_gpu(x_flat, "x_flat", torch.bfloat16)
    rows, slack = batch * (h + 2) * (w + 2), w + 3
    cout, cin = dz_img.shape[-1], x_flat.shape[-1]
    if dz_img.numel() != rows * cout or x_flat.numel() != (rows + 2 * slack) * cin:
        raise ScError(f"conv3x3_dw_bf16: dz_img has {dz_img.numel()} elements, x_flat {x_flat.numel()}; expected {rows} x {cout} and "
                      f"{rows + 2 * slack} x {cin} (halo_buffer)")
    if out is None:
        out = torch.zeros(cout, 9 * cin, dtype=torch.float32, device=dz_img.device)
    elif tuple(out.shape) != (cout, 9 * cin):
        raise ScError(f"conv3x3_dw_bf16: out is {list(out.shape)}, expected [{cout}, {9 * cin}]")
    ws = _workspace(LIB.raw("sc_gemm_bf16_tn_workspace_bytes")(cout, 9 * cin, rows), dz_img.device)
    LIB.call("sc_conv3x3_dw_bf16", ptr(dz_img), ptr(x_flat[slack:]), ptr(out), batch, h, w, cout, cin, 1.0, float(beta), ptr(ws), ws.numel(), stream_ptr())
    return out
