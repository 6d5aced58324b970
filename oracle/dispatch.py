"""Oracle: loss_type dispatch of the training loop (test infrastructure, see oracle/__init__.py).

Restates reference sparsify_clip.py:778-938 as a table: each loss_type string maps
to (has warm-up phase, uniformity flavour, beta-weighted?, alpha-weighted?, lalign?).
First match wins, so the duplicated "EXP 8" string (:833) resolves to the EXP-7 row
(:813) exactly as the reference's if/elif chain does.
"""
from __future__ import annotations

from . import loss_head as L
from .schedules import get_alpha, get_beta

# (loss_type, warmup_phase, unif in {"none","both","centroids"}, use_beta, use_alpha, use_lalign)  -- reference line
TABLE = [
    ("anchor", False, "none", False, False, False),                                                          # :778
    ("only_lunif_n_then_anchor+lalign+lunif(text)+lunif(img)", True, "both", False, False, True),           # :782
    ("only_lunif_n_then_anchor+lalign+lunif(centroids)", True, "centroids", False, False, True),            # :794
    ("only_lunif_n_then_anchor+lalign+BETA*lunif(centroids)", True, "both", True, False, True),             # :813 (EXP 7 math; :833 is dead)
    ("only_lunif_n_then_anchor+ALPHA*lalign+BETA*(lunif(text)+lunif(img))", True, "both", True, True, True),  # :854
    ("only_lunif_n_then_anchor+ALPHA*lalign+BETA*lunif(centroids)", True, "centroids", True, True, True),   # :879
    ("ANCHOR(IMAGE,TEXT)+LALIGN(IMAGE,TEXT)+LUNIF(CENTROIDS)", False, "centroids", False, False, True),     # :909
    ("ANCHOR(IMAGE,TEXT)+LALIGN(IMAGE,TEXT)", False, "none", False, False, True),                           # :922
    ("ANCHOR(IMAGE,TEXT)+LUNIF(CENTROIDS)", False, "centroids", False, False, False),                       # :930
]


def lookup(loss_type):
    for row in TABLE:
        if row[0] == loss_type:
            return row
    raise KeyError(loss_type)


def compose_loss(config, image_embeds, text_embeds, temperature, epoch, current_batch, t_total):
    """Returns (loss, beta, alpha) for one step; beta/alpha are None when the branch leaves them untouched."""
    _, warm, unif, use_beta, use_alpha, use_lalign = lookup(config["loss_type"])
    if warm and epoch < config["only_lunif_epochs"]:
        # warm-up phase: (lunif(img)+lunif(txt))/2, e.g. :783-786
        return (L.lunif_loss(image_embeds) + L.lunif_loss(text_embeds)) / 2, None, None
    loss = L.contrastive_loss(image_embeds, text_embeds, temperature=temperature)
    beta = alpha = None
    if use_lalign:
        la = L.lalign_loss(image_embeds, text_embeds)
        if use_alpha:
            alpha = get_alpha(current_batch, t_total, config["alpha_warmup_epoch"], config["alpha_increment_epoch"])
            la = alpha * la
        loss = loss + la
    if unif != "none":
        if unif == "both":
            lu = (L.lunif_loss(image_embeds) + L.lunif_loss(text_embeds)) / 2
        else:
            lu = L.lunif_centroids(image_embeds, text_embeds)
        if use_beta:
            beta = get_beta(current_batch, t_total, config["beta_warmup_epoch"], config["beta_decay_epoch"])
            lu = beta * lu
        loss = loss + lu
    return loss, beta, alpha
