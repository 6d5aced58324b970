"""Oracle: loss head of the training step (test infrastructure, see oracle/__init__.py).

Every function is a CPU restatement of the cited reference lines and works on
torch CPU tensors of any float dtype (fp32 = the reference CPU path, fp64 = a
high-precision yardstick).  The ``*_grads`` helpers are closed-form fp64
gradients used to check the HIP backward kernels without autograd.
"""
from __future__ import annotations

import numpy as np
import torch


# --------------------------------------------------------------------------- #
# forward restatements (differentiable through torch autograd)
# --------------------------------------------------------------------------- #
def normalize_rows(x: torch.Tensor) -> torch.Tensor:
    """x / ||x||_2 per row, no epsilon.  Reference sparsify_clip.py:772-773."""
    return x / x.norm(dim=-1, keepdim=True)


def normalize_rows_eps(x: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """F.normalize(x, dim=-1): x / max(||x||, eps).  Reference sparsify_clip.py:804."""
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


def contrastive_loss(image_embeds, text_embeds, temperature=0.07):
    """Symmetric InfoNCE with a divisor temperature.  Reference sparsify_clip.py:110-132.

    logits = I @ T^T / temperature; mean CE over rows and over columns against
    the diagonal; average of the two directions.
    """
    logits = (image_embeds @ text_embeds.t()) / temperature
    diag = logits.diagonal()
    row_lse = torch.logsumexp(logits, dim=1)
    col_lse = torch.logsumexp(logits, dim=0)
    return ((row_lse - diag).mean() + (col_lse - diag).mean()) / 2


def contrastive_loss_soft(image_embeds, text_embeds, soft_targets, temperature=0.07):
    """Soft-target variant.  Reference sparsify_clip.py:135-157 (contrastive_loss_roberta)."""
    logits = (image_embeds @ text_embeds.t()) / temperature
    li = -(soft_targets * torch.log_softmax(logits, dim=1)).sum(dim=1).mean()
    lt = -(soft_targets.t() * torch.log_softmax(logits.t(), dim=1)).sum(dim=1).mean()
    return (li + lt) / 2


def lunif_loss(x, t=2):
    """log mean_{i<j} exp(-t ||x_i - x_j||^2).  Reference sparsify_clip.py:159-164.

    Restated with explicit differences over the strict upper triangle (what
    torch.pdist enumerates) instead of calling pdist.
    """
    n = x.shape[0]
    total = x.new_zeros(())
    rows = max(1, min(n, (1 << 25) // max(1, n * x.shape[1])))  # bound the diff block to ~128 MB fp32
    for i0 in range(0, n - 1, rows):
        i1 = min(n - 1, i0 + rows)
        diff = x[i0:i1, None, :] - x[None, i0 + 1:, :]            # [r, n-i0-1, d]
        keep = torch.arange(i0 + 1, n)[None, :] > torch.arange(i0, i1)[:, None]
        ss = torch.where(keep, diff.pow(2).sum(dim=2), torch.ones((), dtype=x.dtype))  # j <= i never reaches sqrt
        sq = ss.sqrt().pow(2)                                    # pdist returns the norm; the reference squares it
        total = total + (sq.mul(-t).exp() * keep).sum()
    return (total / (n * (n - 1) // 2)).log()


def lunif_loss_gram(x, t=2):
    """Same quantity through the Gram matrix (the formulation the HIP kernel uses):
    d_ij = |x_i|^2 + |x_j|^2 - 2 x_i.x_j clamped at 0, mean over i != j."""
    n = x.shape[0]
    g = x @ x.t()
    nrm = (x * x).sum(dim=1)
    d = (nrm[:, None] + nrm[None, :] - 2 * g).clamp_min(0)
    w = torch.exp(-t * d)
    w = w - torch.diag(torch.diagonal(w))
    return (w.sum() / (n * (n - 1))).log()


def lalign_loss(x, y, alpha=2):
    """mean_i ||x_i - y_i||_2^alpha.  Reference sparsify_clip.py:186-187."""
    return (x - y).norm(dim=1).pow(alpha).mean()


def compute_centroids_only(a, b):
    """Pairwise midpoint (a+b)/2.  Reference sparsify_clip.py:334-355."""
    return (a + b) / 2.0


def compute_centroids(text_embeddings, visual_embeddings):
    """All-pairs midpoints and their norms.  Reference sparsify_clip.py:308-332."""
    c = (text_embeddings[:, None, :] + visual_embeddings[None, :, :]) / 2.0
    return c.norm(dim=-1), c


def sparsify_loss(x):
    """mse(x x^T, 2I-1).  Reference sparsify_clip.py:166-176."""
    g = x @ x.t()
    n = g.shape[0]
    target = 2 * torch.eye(n, dtype=g.dtype) - 1
    return (g - target).pow(2).mean()


def centroid_alignment_loss(img, txt, p=2):
    """|| mean(img) - mean(txt) ||_p.  Reference sparsify_clip.py:487-505."""
    return torch.norm(img.mean(dim=0) - txt.mean(dim=0), p=p)


def lunif_centroids(image_embeds, text_embeds, t=2):
    """lunif of the eps-normalised pair midpoints.  Reference sparsify_clip.py:803-805."""
    return lunif_loss(normalize_rows_eps(compute_centroids_only(image_embeds, text_embeds)), t)


# --------------------------------------------------------------------------- #
# closed-form fp64 gradients (numpy)
# --------------------------------------------------------------------------- #
def _lse(a, axis):
    m = a.max(axis=axis, keepdims=True)
    return (m + np.log(np.exp(a - m).sum(axis=axis, keepdims=True))).squeeze(axis)


def contrastive_grads(img, txt, temperature):
    """Returns (loss, dI, dT, dtemperature) in fp64 for the loss of sparsify_clip.py:110-132."""
    i64, t64 = np.asarray(img, np.float64), np.asarray(txt, np.float64)
    b = i64.shape[0]
    logits = i64 @ t64.T / temperature
    r, c = _lse(logits, 1), _lse(logits, 0)
    diag = np.diag(logits)
    loss = ((r - diag).mean() + (c - diag).mean()) / 2
    g = (np.exp(logits - r[:, None]) + np.exp(logits - c[None, :])) / (2 * b)
    g[np.arange(b), np.arange(b)] -= 1.0 / b
    d_img = g @ t64 / temperature
    d_txt = g.T @ i64 / temperature
    d_temp = -(g * logits).sum() / temperature
    return loss, d_img, d_txt, d_temp


def lunif_grads(x, t=2):
    """Returns (loss, dX) in fp64 for sparsify_clip.py:159-164."""
    x64 = np.asarray(x, np.float64)
    n = x64.shape[0]
    nrm = (x64 * x64).sum(1)
    d = np.maximum(nrm[:, None] + nrm[None, :] - 2 * x64 @ x64.T, 0)
    w = np.exp(-t * d)
    np.fill_diagonal(w, 0.0)
    s_all = w.sum()
    loss = np.log(s_all / (n * (n - 1)))
    s_row = w.sum(1)
    dx = (-4.0 * t / s_all) * (s_row[:, None] * x64 - w @ x64)
    return loss, dx


def lalign_grads(x, y, alpha=2):
    """Returns (loss, dX, dY) in fp64 for sparsify_clip.py:186-187 (sub-gradient 0 at zero distance)."""
    x64, y64 = np.asarray(x, np.float64), np.asarray(y, np.float64)
    diff = x64 - y64
    nrm = np.sqrt((diff * diff).sum(1))
    loss = (nrm ** alpha).mean()
    with np.errstate(divide="ignore", invalid="ignore"):
        coef = np.where(nrm > 0, alpha * nrm ** (alpha - 2), 0.0) / x64.shape[0]
    dx = coef[:, None] * diff
    return loss, dx, -dx


def sparsify_grads(x):
    """Returns (loss, dX) in fp64 for sparsify_clip.py:166-176."""
    x64 = np.asarray(x, np.float64)
    n = x64.shape[0]
    dmat = x64 @ x64.T - (2 * np.eye(n) - 1)
    return (dmat ** 2).mean(), (4.0 / (n * n)) * dmat @ x64


def normalize_backward(x, dy, eps=None):
    """Backward of row normalisation y = x/max(|x|,eps) in fp64 (eps=None: no clamp)."""
    x64, dy64 = np.asarray(x, np.float64), np.asarray(dy, np.float64)
    nrm = np.sqrt((x64 * x64).sum(1, keepdims=True))
    if eps is not None:
        clamped = nrm < eps
        nrm = np.maximum(nrm, eps)
    y = x64 / nrm
    dx = (dy64 - y * (dy64 * y).sum(1, keepdims=True)) / nrm
    if eps is not None:
        dx = np.where(clamped, dy64 / nrm, dx)
    return dx


# --------------------------------------------------------------------------- #
# synthetic embedding sets shared by fixtures, tests and bench (build-owned Philox stream)
# --------------------------------------------------------------------------- #
def philox_embeddings(seed: int, b: int, d: int, clustered: bool = False):
    """Deterministic unit-norm fp32 embedding pair (img, txt) from numpy Philox.

    ``clustered``: 64 centres + 0.1*noise, renormalised - the
    near-duplicate case where 2-2G cancels.
    """
    rng = np.random.Generator(np.random.Philox(seed))

    def one():
        if clustered:
            centres = rng.standard_normal((64, d), dtype=np.float32)
            centres /= np.linalg.norm(centres, axis=1, keepdims=True)
            pick = rng.integers(0, 64, size=b)
            z = centres[pick] + 0.1 * rng.standard_normal((b, d), dtype=np.float32) / np.sqrt(d)
        else:
            z = rng.standard_normal((b, d), dtype=np.float32)
        z = z.astype(np.float32)
        z /= np.linalg.norm(z, axis=1, keepdims=True)
        return z.astype(np.float32)

    return one(), one()
