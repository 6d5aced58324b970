"""Oracle: eval-time metrics (test infrastructure, see oracle/__init__.py).

Retrieval ranks and geometry metrics of reference sparsify_clip.py:357-528 and the five
Wasserstein-uniformity variants of reference uniformity.py:6-205.
"""
from __future__ import annotations

import math

import numpy as np
import torch


# ----------------------------------------------------------------------------- retrieval
def retrieval_ranks(score: torch.Tensor, direction: str) -> torch.Tensor:
    """Rank of the ground-truth item per query when ids are range(N) on both sides.

    Reference sparsify_clip.py:372-380 (forward: sort rows descending, position of i in row i)
    and :394-402 (backward: sort columns).  torch.sort is not stable by default, so the
    rank under exact ties is whatever the sort returns; we reproduce by sorting too.
    """
    if direction == "forward":
        order = score.sort(dim=-1, descending=True)[1]
    else:
        order = score.sort(dim=0, descending=True)[1].t()
    n = order.shape[0]
    return (order == torch.arange(n)[:, None]).float().argmax(dim=1)


def retrieval_metrics(score: torch.Tensor, direction: str) -> dict:
    """R@1/5/10 and their mean as percentages rounded to 4 dp.  Reference :382-392, :404-414."""
    rank = retrieval_ranks(score, direction)
    n = rank.numel()
    r = [(rank < k).sum().item() / n for k in (1, 5, 10)]
    p = "forward" if direction == "forward" else "backward"
    return {f"{p}_r1": round(r[0] * 100, 4), f"{p}_r5": round(r[1] * 100, 4),
            f"{p}_r10": round(r[2] * 100, 4), f"{p}_ravg": round(sum(r) / 3 * 100, 4)}


def compute_gap(a, b):
    """|| mean(a) - mean(b) ||.  Reference sparsify_clip.py:418-436."""
    return torch.norm(a.mean(dim=0) - b.mean(dim=0)).item()


def mean_angular_value(x):
    """Mean off-diagonal cosine.  Reference sparsify_clip.py:438-457."""
    g = x @ x.t()
    n = g.shape[0]
    return ((g.sum() - g.diagonal().sum()) / (n * (n - 1))).item()


def mean_cosine_true_pairs(a, b):
    """Mean of diag(a b^T).  Reference sparsify_clip.py:508-528."""
    return (a * b).sum(dim=1).mean().item()


# ----------------------------------------------------------------------------- uniformity
def _moments(x):
    n = x.shape[0]
    mu = x.mean(dim=0, keepdim=True)
    xc = x - mu
    return mu, xc.t() @ xc / n


def _w2(mu_sq, m, tr_sigma, tr_sqrt):
    return mu_sq + 1 + tr_sigma - (2.0 / math.sqrt(m)) * tr_sqrt


def torch_uniformity1(f1):
    """SVD of Sigma, returns +W2.  Reference uniformity.py:6-51."""
    mu, sigma = _moments(f1)
    tr = torch.clamp(torch.trace(sigma), min=0)
    u, s, _ = torch.linalg.svd(sigma)
    root = u @ torch.diag(torch.sqrt(torch.clamp(s + 1e-8, min=0))) @ u.T
    return torch.sqrt(_w2(torch.norm(mu) ** 2, f1.shape[1], tr, torch.trace(root)))


def torch_uniformity(f1, f2):
    """Concat, +1e-6 on every entry of Sigma, eigh, returns -W2.  Reference uniformity.py:53-98."""
    x = torch.cat([f1, f2], dim=0)
    mu, sigma = _moments(x)
    sigma = sigma + 1e-6
    w, v = torch.linalg.eigh(sigma)
    root = v @ torch.diag(torch.sqrt(torch.clamp(w + 1e-8, min=0))) @ v.T
    return -torch.sqrt(_w2(torch.norm(mu) ** 2, x.shape[1], torch.trace(sigma), torch.trace(root)))


def numpy_uniformity(f1, f2):
    """np.linalg.eig, returns python float -W2.  Reference uniformity.py:101-128
    (the reference also prints covariance.shape at :108; the oracle does not) and
    sparsify_clip.py:459-485 (same arithmetic without the print)."""
    x = torch.cat([f1, f2], dim=0)
    mu, sigma = _moments(x)
    cov = sigma.detach().cpu().numpy()
    mean = x.mean(0).detach().cpu().numpy()
    s, q = np.linalg.eig(cov)
    root = q @ np.sqrt(np.diag((s + 1e-8).clip(min=0))) @ q.T
    part2 = np.trace(cov - 2.0 / np.sqrt(x.shape[1]) * root)
    return -math.sqrt(np.sum(mean * mean) + 1 + part2)


def torch_uniformity_equivalent(f1):
    """torch.linalg.eig real parts, returns +W2.  Reference uniformity.py:138-180."""
    mu, sigma = _moments(f1)
    w, v = torch.linalg.eig(sigma)
    w, v = w.real + 1e-8, v.real
    root = v @ torch.sqrt(torch.diag(torch.clamp(w, min=0))) @ v.t()
    mean = f1.mean(0)
    part2 = torch.trace(sigma - 2.0 / math.sqrt(f1.shape[1]) * root)
    return torch.sqrt(torch.sum(mean * mean) + 1 + part2)


def uniformity10(z1):
    """abs of eigenvalues AND eigenvectors, returns +W2.  Reference uniformity.py:182-205."""
    mu, sigma = _moments(z1)
    s, q = torch.linalg.eig(sigma)
    s, q = torch.abs(s), torch.abs(q)
    root = q @ torch.sqrt(torch.diag(s)) @ q.T
    mean = z1.mean(0)
    part2 = torch.trace(sigma - 2.0 / math.sqrt(z1.shape[1]) * root)
    return torch.sqrt(torch.sum(mean * mean) + 1 + part2)
