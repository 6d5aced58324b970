"""Oracle: host-side scalar schedules (test infrastructure, see oracle/__init__.py)."""
from __future__ import annotations

import math


def _ramp(step, total_steps, hold_epochs, ramp_epochs):
    """Fraction of the linear ramp completed at `step`; "epoch" = total_steps/100.

    Shared core of reference sparsify_clip.py:41-51 (get_beta) and :54-64 (get_alpha):
    0 during the hold, rises linearly over `ramp_epochs`, then None once past it.
    """
    per_epoch = total_steps / 100
    if step < hold_epochs * per_epoch:
        return 0.0
    if step < (hold_epochs + ramp_epochs) * per_epoch:
        return float(step - hold_epochs * per_epoch) / float(max(1, ramp_epochs * per_epoch))
    return None


def get_beta(current_step, total_steps, warmup_epoch=20, decay_epoch=50):
    """1.0 -> linear decay -> 0.0.  Reference sparsify_clip.py:41-51."""
    f = _ramp(current_step, total_steps, warmup_epoch, decay_epoch)
    return 0.0 if f is None else 1.0 - f


def get_alpha(current_step, total_steps, warmup_epoch=20, increment_epoch=50):
    """1.0 -> linear rise -> 2.0.  Reference sparsify_clip.py:54-64."""
    f = _ramp(current_step, total_steps, warmup_epoch, increment_epoch)
    return 2.0 if f is None else 1.0 + f


def lr_multiplier(step, num_warmup_steps, num_training_steps, only_lunif_epochs,
                  num_cycles=0.5, steps_sparsify=462):
    """LambdaLR multiplier of reference sparsify_clip.py:97-105.

    Holds 1.0 while step < steps_sparsify iff only_lunif_epochs > 0; else linear
    warm-up then half-cosine decay.
    """
    if step < steps_sparsify and only_lunif_epochs > 0:
        return 1.0
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))
