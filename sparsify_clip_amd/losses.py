"""Loss head with the reference's function signatures (sparsify_clip.py:110-187, :308-355, :487-505).

Every function takes [B,D] fp32 CUDA tensors and returns a 0-dim differentiable tensor, exactly like the
reference's module-level functions, but the arithmetic is the HIP library's fused forward+backward kernels
(the gradient is produced together with the value and handed to autograd in ``backward``).
Temperature may be a python float or a 0-dim tensor / nn.Parameter (learnable, :716-717); in the reference the
learnable temperature lives on the CPU, so its value is read on the host here as well.
"""
from __future__ import annotations

import torch

from . import ops
from ._lib import ScError


def _as_f32(x, name):
    if not isinstance(x, torch.Tensor) or x.dim() != 2:
        raise ScError(f"{name} must be a [B,D] tensor")
    if not x.is_cuda:
        raise ScError(f"{name} is on {x.device}: the loss head has no CPU path (the HIP extension does the work)")
    return x.contiguous().float()


class _Contrastive(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, txt, temp_tensor, temp_value):
        need = img.requires_grad or txt.requires_grad or (temp_tensor is not None and temp_tensor.requires_grad)
        loss, d_img, d_txt, d_temp = ops.contrastive_fwd_bwd(img, txt, temp_value, need_grad=need, need_dtemp=temp_tensor is not None)
        ctx.save_for_backward(d_img, d_txt, d_temp)
        ctx.temp_device = temp_tensor.device if temp_tensor is not None else None
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        d_img, d_txt, d_temp = ctx.saved_tensors
        gt = None
        if d_temp is not None:
            gt = (d_temp.reshape(()) * g).to(ctx.temp_device)
        return d_img * g, d_txt * g, gt, None


def contrastive_loss(image_embeds, text_embeds, temperature=0.07):
    """Symmetric InfoNCE, logits / temperature.  Reference :110-132."""
    img, txt = _as_f32(image_embeds, "image_embeds"), _as_f32(text_embeds, "text_embeds")
    if img.shape != txt.shape:
        raise ScError(f"image_embeds {tuple(img.shape)} and text_embeds {tuple(txt.shape)} differ")
    if isinstance(temperature, torch.Tensor):
        return _Contrastive.apply(img, txt, temperature, float(temperature.detach()))
    return _Contrastive.apply(img, txt, None, float(temperature))


class _Lunif(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, t):
        loss, dx = ops.lunif_fwd_bwd(x, t, need_grad=x.requires_grad)
        ctx.save_for_backward(dx)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return dx * g, None


def lunif_loss(x, t=2):
    """log mean_{i<j} exp(-t ||xi - xj||^2).  Reference :159-164."""
    return _Lunif.apply(_as_f32(x, "x"), float(t))


class _Lalign(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, alpha):
        loss, dx, dy = ops.lalign_fwd_bwd(x, y, alpha, need_grad=x.requires_grad or y.requires_grad)
        ctx.save_for_backward(dx, dy)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dx, dy = ctx.saved_tensors
        return dx * g, dy * g, None


def lalign_loss(x, y, alpha=2):
    """mean_i ||xi - yi||^alpha.  Reference :186-187."""
    return _Lalign.apply(_as_f32(x, "x"), _as_f32(y, "y"), float(alpha))


def random_alignment_loss(x, y):
    """lalign against a random permutation of y (host RNG, as the reference's torch.randperm).  Reference :178-184."""
    idx = torch.randperm(y.size(0)).to(y.device)
    return lalign_loss(x, y[idx], alpha=2)


class _Sparsify(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        loss, dx = ops.sparsify_fwd_bwd(x, need_grad=x.requires_grad)
        ctx.save_for_backward(dx)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return (dx * g,)


def sparsify_loss(x):
    """mse(x x^T, 2I - 1).  Reference :166-176 (defined there, used by no shipped YAML)."""
    return _Sparsify.apply(_as_f32(x, "x"))


class _CentroidNormalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        c, inv = ops.centroid_fwd(a, b)
        ctx.save_for_backward(c, inv)
        return c

    @staticmethod
    def backward(ctx, dc):
        c, inv = ctx.saved_tensors
        da, db = torch.zeros_like(c), torch.zeros_like(c)
        ops.centroid_bwd_accumulate(c, inv, dc.contiguous().float(), da, db)
        return da, db


def normalized_centroids(image_embeds, text_embeds):
    """F.normalize(compute_centroids_only(a, b), dim=-1) in one kernel.  Reference :803-804."""
    return _CentroidNormalize.apply(_as_f32(image_embeds, "image_embeds"), _as_f32(text_embeds, "text_embeds"))


def compute_centroids_only(text_embeddings, visual_embeddings):
    """(a + b) / 2 row by row.  Reference :334-355.  (Pure index-free elementwise op on a [B,D] tensor.)"""
    return (text_embeddings + visual_embeddings) / 2.0


def compute_centroids(text_embeddings, visual_embeddings):
    """All-pairs midpoints [B1,B2,D] and their norms.  Reference :308-332 (O(B^2 D) memory; unused by any config)."""
    c = (text_embeddings.unsqueeze(1) + visual_embeddings.unsqueeze(0)) / 2.0
    return torch.norm(c, dim=-1), c


def centroid_alignment_loss(img_embeds, txt_embeds, p=2):
    """|| mean(img) - mean(txt) ||_p.  Reference :487-505 (unused by any config; thin torch composition)."""
    return torch.norm(img_embeds.mean(dim=0) - txt_embeds.mean(dim=0), p=p)


def contrastive_loss_roberta(image_embeds, text_embeds, roberta_similarity, temperature=0.07):
    """Soft-target InfoNCE.  Reference :135-157; reachable only through loss_type 'anchor-roberta', which needs a
    model-name fetch (:713) and is used by no YAML - kept for signature parity as a thin torch composition."""
    logits = (image_embeds @ text_embeds.t()) / temperature
    li = torch.nn.functional.cross_entropy(logits, roberta_similarity)
    lt = torch.nn.functional.cross_entropy(logits.t(), roberta_similarity.t())
    return (li + lt) / 2
