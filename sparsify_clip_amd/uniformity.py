"""The five Wasserstein-uniformity metrics of reference uniformity.py (:6-205), same names, argument lists and
return-sign conventions, and the eval-time geometry metrics of sparsify_clip.py:418-528.

Common core: mu = mean(x), Sigma = (x-mu)^T (x-mu) / N,
W2 = sqrt(|mu|^2 + 1 + tr Sigma - 2/sqrt(m) tr Sigma^(1/2)).
On a GPU tensor the [N,m]^T [N,m] covariance contraction runs on the fp32 MFMA GEMM of the HIP library; the m x m
eigen-decompositions use torch.linalg / numpy exactly as the reference does (they are not on the training path).
CPU tensors are accepted too (pure torch), since these are offline analysis functions in the reference.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def _moments(x):
    n = x.size(0)
    mu = torch.mean(x, dim=0, keepdim=True)
    xc = (x - mu).contiguous()
    if xc.is_cuda and xc.dtype == torch.float32:
        from . import ops
        cov = ops.gemm_f32(xc, xc, trans_a=True) / n
    else:
        cov = torch.mm(xc.t(), xc) / n
    return mu, cov


def _sqrtm_trace_sym(vals, vecs):
    root = vecs @ torch.diag(torch.sqrt(torch.clamp(vals, min=0))) @ vecs.T
    return torch.trace(root)


def torch_uniformity1(features_modality1):
    """SVD of Sigma; returns +W2.  Reference uniformity.py:6-51."""
    x = features_modality1
    mu, sigma = _moments(x)
    tr = torch.clamp(torch.trace(sigma), min=0)
    u, s, _ = torch.linalg.svd(sigma)
    tr_root = _sqrtm_trace_sym(s + 1e-8, u)
    m = x.shape[1]
    return torch.sqrt(torch.norm(mu) ** 2 + 1 + tr - (2 / torch.sqrt(torch.tensor(m, dtype=sigma.dtype))) * tr_root)


def torch_uniformity(features_modality1, features_modality2):
    """Both modalities concatenated, 1e-6 added to every entry of Sigma, eigh; returns -W2.  Reference :53-98."""
    x = torch.cat([features_modality1, features_modality2], dim=0)
    mu, sigma = _moments(x)
    sigma = sigma + 1e-6
    vals, vecs = torch.linalg.eigh(sigma)
    tr_root = _sqrtm_trace_sym(vals + 1e-8, vecs)
    m = x.shape[1]
    w2 = torch.sqrt(torch.norm(mu) ** 2 + 1 + torch.trace(sigma) - (2 / torch.sqrt(torch.tensor(m, dtype=sigma.dtype))) * tr_root)
    return -w2


def numpy_uniformity(features_modality1, features_modality2):
    """numpy eig on the host; returns a python float -W2.  Reference :101-128 (which also prints covariance.shape at
    :108 - library code here does not print)."""
    x = torch.cat([features_modality1, features_modality2], dim=0)
    _, sigma = _moments(x)
    cov = sigma.detach().cpu().numpy()
    mean = x.mean(0).detach().cpu().numpy()
    vals, vecs = np.linalg.eig(cov)
    root = np.dot(np.dot(vecs, np.sqrt(np.diag((vals + 1e-8).clip(min=0)))), vecs.T)
    part2 = np.trace(cov - 2.0 / np.sqrt(x.size(1)) * root)
    return -math.sqrt(np.sum(mean * mean) + 1 + part2)


def torch_uniformity_equivalent(features_modality1):
    """torch.linalg.eig, real parts; returns +W2.  Reference :138-180."""
    x = features_modality1
    _, sigma = _moments(x)
    vals, vecs = torch.linalg.eig(sigma)
    vals, vecs = vals.real + 1e-8, vecs.real
    root = torch.mm(torch.mm(vecs, torch.sqrt(torch.diag(torch.clamp(vals, min=0)))), vecs.t())
    mean = x.mean(0)
    part2 = torch.trace(sigma - 2.0 / math.sqrt(x.size(1)) * root)
    return torch.sqrt(torch.sum(mean * mean) + 1 + part2)


def uniformity10(z1):
    """abs() of eigenvalues AND eigenvectors; returns +W2.  Reference :182-205."""
    _, sigma = _moments(z1)
    vals, vecs = torch.linalg.eig(sigma)
    vals, vecs = torch.abs(vals), torch.abs(vecs)
    root = torch.mm(torch.mm(vecs, torch.sqrt(torch.diag(vals))), vecs.T)
    mean = z1.mean(0)
    part2 = torch.trace(sigma - 2.0 / math.sqrt(z1.size(1)) * root)
    return torch.sqrt(torch.sum(mean * mean) + 1 + part2)


# ------------------------------------------------------------------------------- sparsify_clip.py eval metrics
def uniformity(features_modality1, features_modality2):
    """sparsify_clip.py:459-485 - numpy_uniformity without the print."""
    return numpy_uniformity(features_modality1, features_modality2)


def _device_metrics(a, b):
    """GPU fp32 inputs: all geometry metrics from ONE library call (sc_eval_metrics), else None."""
    if a.is_cuda and b.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 2 and a.shape == b.shape and a.shape[0] >= 2:
        from . import ops
        return ops.eval_metrics(a.contiguous(), b.contiguous())
    return None


def compute_gap(feat_modality1, feat_modality2):
    """sparsify_clip.py:418-436."""
    m = _device_metrics(feat_modality1, feat_modality2)
    if m is not None:
        return m[0].item()
    return torch.norm(feat_modality1.mean(dim=0) - feat_modality2.mean(dim=0)).item()


def compute_mean_angular_value_of_a_modality(feat_modality):
    """Mean off-diagonal cosine.  sparsify_clip.py:438-457 (the reference builds the [N,N] Gram matrix; the sum of all its entries is
    |sum_i x_i|^2, so the device path needs O(N E) work and no matrix)."""
    m = _device_metrics(feat_modality, feat_modality)
    if m is not None:
        return m[1].item()
    g = feat_modality @ feat_modality.T
    n = g.size(0)
    return ((g.sum() - torch.diagonal(g).sum()) / (n * (n - 1))).item()


def mean_distance_of_true_pairs(features_modality1, features_modality2):
    """Mean cosine of matching pairs.  sparsify_clip.py:508-528."""
    m = _device_metrics(features_modality1, features_modality2)
    if m is not None:
        return m[3].item()
    return (features_modality1 * features_modality2).sum(dim=1).mean().item()
