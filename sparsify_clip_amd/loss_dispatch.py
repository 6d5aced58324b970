"""loss_type dispatch of the training step (reference sparsify_clip.py:778-938) on fused HIP kernels.

The reference composes the step loss from contrastive / lalign / lunif terms with an if/elif chain over nine
strings and lets autograd differentiate it.  Here each string maps to a row of LOSS_TABLE; the value AND the
gradients w.r.t. the (global-batch, L2-normalised) embeddings come straight from the fused forward+backward
kernels, with the term weights (1, beta, alpha, 1/2) passed as the kernels' grad_scale - no autograd graph and
no host synchronisation (the loss stays a device scalar; reference :944 syncs every step).

First match wins, so the duplicated "EXP 8" string (:833) resolves to the EXP-7 arithmetic (:813-829), as in the
reference.  An unknown loss_type raises up front (the reference dies later with AttributeError at :944).
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from . import ops
from ._lib import ScError
from .schedules import get_alpha, get_beta


@dataclass(frozen=True)
class LossSpec:
    warmup_phase: bool     # has the `epoch < only_lunif_epochs` branch
    unif: str              # "none" | "both" (lunif(img)+lunif(txt))/2 | "centroids"
    use_beta: bool
    use_alpha: bool
    use_lalign: bool
    ref_line: int


LOSS_TABLE = {
    "anchor": LossSpec(False, "none", False, False, False, 778),
    "only_lunif_n_then_anchor+lalign+lunif(text)+lunif(img)": LossSpec(True, "both", False, False, True, 782),
    "only_lunif_n_then_anchor+lalign+lunif(centroids)": LossSpec(True, "centroids", False, False, True, 794),
    "only_lunif_n_then_anchor+lalign+BETA*lunif(centroids)": LossSpec(True, "both", True, False, True, 813),
    "only_lunif_n_then_anchor+ALPHA*lalign+BETA*(lunif(text)+lunif(img))": LossSpec(True, "both", True, True, True, 854),
    "only_lunif_n_then_anchor+ALPHA*lalign+BETA*lunif(centroids)": LossSpec(True, "centroids", True, True, True, 879),
    "ANCHOR(IMAGE,TEXT)+LALIGN(IMAGE,TEXT)+LUNIF(CENTROIDS)": LossSpec(False, "centroids", False, False, True, 909),
    "ANCHOR(IMAGE,TEXT)+LALIGN(IMAGE,TEXT)": LossSpec(False, "none", False, False, True, 922),
    "ANCHOR(IMAGE,TEXT)+LUNIF(CENTROIDS)": LossSpec(False, "centroids", False, False, False, 930),
}


def validate_loss_type(loss_type: str) -> LossSpec:
    spec = LOSS_TABLE.get(loss_type)
    if spec is None:
        raise ScError(f"unknown loss_type {loss_type!r}; known: {sorted(LOSS_TABLE)}")
    return spec


@dataclass
class StepLoss:
    loss: torch.Tensor          # [1] fp32 on device
    d_img: torch.Tensor         # [B,E] gradient w.r.t. normalised image embeddings
    d_txt: torch.Tensor
    d_temp: torch.Tensor | None  # [1] gradient w.r.t. the temperature, or None when this branch does not use it
    beta: float | None
    alpha: float | None
    terms: dict


def step_loss(config, image_embeds, text_embeds, temperature, epoch, current_batch, t_total, want_dtemp=False) -> StepLoss:
    """One training step's loss and embedding gradients for config["loss_type"]."""
    spec = validate_loss_type(config["loss_type"])
    dev = image_embeds.device
    total = torch.zeros(1, dtype=torch.float32, device=dev)
    terms = {}
    if spec.warmup_phase and epoch < config["only_lunif_epochs"]:
        # (lunif(img) + lunif(txt)) / 2   e.g. :783-786
        li, d_img = ops.lunif_fwd_bwd(image_embeds, 2.0, grad_scale=0.5)
        lt, d_txt = ops.lunif_fwd_bwd(text_embeds, 2.0, grad_scale=0.5)
        ops.axpy_(total, 0.5, li)
        ops.axpy_(total, 0.5, lt)
        terms.update(lunif_img=li, lunif_txt=lt)
        return StepLoss(total, d_img, d_txt, None, None, None, terms)

    anchor, d_img, d_txt, d_temp = ops.contrastive_fwd_bwd(image_embeds, text_embeds, temperature, need_dtemp=want_dtemp)
    ops.axpy_(total, 1.0, anchor)
    terms["anchor"] = anchor
    beta = alpha = None
    if spec.use_lalign:
        w = 1.0
        if spec.use_alpha:
            alpha = w = get_alpha(current_batch, t_total, config["alpha_warmup_epoch"], config["alpha_increment_epoch"])
        la, dx, dy = ops.lalign_fwd_bwd(image_embeds, text_embeds, 2.0, grad_scale=w)
        ops.axpy_(total, w, la)
        ops.axpy_(d_img, 1.0, dx)
        ops.axpy_(d_txt, 1.0, dy)
        terms["lalign"] = la
    if spec.unif != "none":
        w = 1.0
        if spec.use_beta:
            beta = w = get_beta(current_batch, t_total, config["beta_warmup_epoch"], config["beta_decay_epoch"])
        if spec.unif == "both":
            li, dxi = ops.lunif_fwd_bwd(image_embeds, 2.0, grad_scale=0.5 * w)
            lt, dxt = ops.lunif_fwd_bwd(text_embeds, 2.0, grad_scale=0.5 * w)
            ops.axpy_(total, 0.5 * w, li)
            ops.axpy_(total, 0.5 * w, lt)
            ops.axpy_(d_img, 1.0, dxi)
            ops.axpy_(d_txt, 1.0, dxt)
            terms.update(lunif_img=li, lunif_txt=lt)
        else:
            c, inv = ops.centroid_fwd(image_embeds, text_embeds)          # F.normalize((img+txt)/2)  :803-804
            lc, dc = ops.lunif_fwd_bwd(c, 2.0, grad_scale=w)
            ops.axpy_(total, w, lc)
            ops.centroid_bwd_accumulate(c, inv, dc, d_img, d_txt)
            terms["lunif_centroids"] = lc
    return StepLoss(total, d_img, d_txt, d_temp, beta, alpha, terms)


def step_loss_rows(config, image_embeds, text_embeds, temperature, epoch, current_batch, t_total, row0, rows, exchange, want_dtemp=False) -> StepLoss:
    """step_loss for a rank that owns rows [row0, row0 + rows) of the gathered batch (equal shards, rank-major): the O(B^2) terms
    visit this rank's rows x all columns only (world times less work than the replicated loss head), every rank obtains the same
    loss value, and d_img / d_txt are THIS RANK'S ROWS of the gradients ([rows, E]).  `exchange(packet [P]) -> [world, P]` is the
    one collective: every rank's row / column LSE, diagonal logits and the three scalars the other terms need.  d_temp is this rank's
    part (SUM over ranks gives the gradient).  Same arithmetic per element as step_loss; the value differs from the replicated one
    only by fp32 summation order."""
    spec = validate_loss_type(config["loss_type"])
    dev = image_embeds.device
    b, e = image_embeds.shape
    a, z = row0, row0 + rows
    total = torch.zeros(1, dtype=torch.float32, device=dev)
    terms = {}
    share = rows / b
    warm = spec.warmup_phase and epoch < config["only_lunif_epochs"]
    beta = alpha = None
    w_la = w_un = 1.0
    if not warm:
        if spec.use_lalign and spec.use_alpha:
            alpha = w_la = get_alpha(current_batch, t_total, config["alpha_warmup_epoch"], config["alpha_increment_epoch"])
        if spec.unif != "none" and spec.use_beta:
            beta = w_un = get_beta(current_batch, t_total, config["beta_warmup_epoch"], config["beta_decay_epoch"])
    # ---- phase 1: this rank's statistics
    packet = torch.zeros(3 * rows + 3, dtype=torch.float32, device=dev)
    unif_inputs = []     # (name, x_all, weight, rowsum, wx, slot in the packet)
    if warm or spec.unif == "both":
        unif_inputs = [("lunif_img", image_embeds, 0.5 * w_un), ("lunif_txt", text_embeds, 0.5 * w_un)]
    elif spec.unif == "centroids":
        c, inv = ops.centroid_fwd(image_embeds, text_embeds)          # all rows: O(B E), every rank needs every centroid as a column
        unif_inputs = [("lunif_centroids", c, w_un)]
    if not warm:
        packet[: 3 * rows].copy_(ops.contrastive_rows_stats(image_embeds, text_embeds, row0, rows, temperature).reshape(-1))
    unif_state = []
    for k, (name, x, w) in enumerate(unif_inputs):
        rowsum, wx, s_part = ops.lunif_rows_stats(x, row0, rows, 2.0)
        packet[3 * rows + k: 3 * rows + k + 1].copy_(s_part)
        unif_state.append((name, x, w, rowsum, wx, 3 * rows + k))
    d_img = d_txt = None
    if not warm and spec.use_lalign:   # a sum over pairs: this rank's pairs, weighted by its share of the batch
        la, d_img, d_txt = ops.lalign_fwd_bwd(image_embeds[a:z], text_embeds[a:z], 2.0, grad_scale=w_la * share)
        packet[3 * rows + 2: 3 * rows + 3].copy_(la)
    # ---- the one collective
    packets = exchange(packet)                       # [world, P]
    world = packets.shape[0]
    # ---- phase 2: loss value (identical on every rank) and this rank's rows of the gradients
    d_temp = None
    if not warm:
        r_all = packets[:, 0:rows].reshape(-1).contiguous()
        c_all = packets[:, rows:2 * rows].reshape(-1).contiguous()
        g_all = packets[:, 2 * rows:3 * rows].reshape(-1).contiguous()
        anchor, gi, gt, d_temp = ops.contrastive_rows_grad(image_embeds, text_embeds, row0, rows, temperature, r_all, c_all, g_all, need_dtemp=want_dtemp)
        ops.axpy_(total, 1.0, anchor)
        terms["anchor"] = anchor
        if d_img is None:
            d_img, d_txt = gi, gt
        else:
            ops.axpy_(d_img, 1.0, gi)
            ops.axpy_(d_txt, 1.0, gt)
        if spec.use_lalign:
            la_all = torch.zeros(1, dtype=torch.float32, device=dev)
            for r in range(world):                   # fixed order: the same value on every rank
                ops.axpy_(la_all, share, packets[r, 3 * rows + 2: 3 * rows + 3])
            ops.axpy_(total, w_la, la_all)
            terms["lalign"] = la_all
    for name, x, w, rowsum, wx, slot in unif_state:
        s_parts = packets[:, slot].contiguous()
        lu, dx = ops.lunif_rows_grad(x[a:z], b, 2.0, w, s_parts, rowsum, wx)
        ops.axpy_(total, w, lu)
        terms[name] = lu
        if name == "lunif_centroids":
            if d_img is None:
                d_img, d_txt = torch.zeros(rows, e, dtype=torch.float32, device=dev), torch.zeros(rows, e, dtype=torch.float32, device=dev)
            ops.centroid_bwd_accumulate(x[a:z], inv[a:z], dx, d_img, d_txt)
        else:
            tgt = "img" if name == "lunif_img" else "txt"
            if tgt == "img":
                d_img = dx if d_img is None else ops.axpy_(d_img, 1.0, dx)
            else:
                d_txt = dx if d_txt is None else ops.axpy_(d_txt, 1.0, dx)
    return StepLoss(total, d_img, d_txt, d_temp if not warm else None, beta, alpha, terms)
