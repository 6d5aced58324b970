"""Host-side scalar schedules with the reference's signatures (sparsify_clip.py:41-107)."""
from __future__ import annotations

import math

from torch.optim.lr_scheduler import LambdaLR


def _phase(current_step, total_steps, hold, ramp):
    """(done_fraction, finished) of the linear ramp; one "epoch" is total_steps/100 steps (reference :43, :56)."""
    per_epoch = total_steps / 100
    if current_step < hold * per_epoch:
        return 0.0, False
    if current_step < (hold + ramp) * per_epoch:
        return float(current_step - hold * per_epoch) / float(max(1, ramp * per_epoch)), False
    return 1.0, True


def get_beta(current_step, total_steps, warmup_epoch=20, decay_epoch=50):
    """L_unif weight: 1.0, then linear decay to 0.0.  Reference :41-51."""
    frac, done = _phase(current_step, total_steps, warmup_epoch, decay_epoch)
    return 0.0 if done else 1.0 - frac


def get_alpha(current_step, total_steps, warmup_epoch=20, increment_epoch=50):
    """L_align weight: 1.0, then linear rise to 2.0.  Reference :54-64."""
    frac, done = _phase(current_step, total_steps, warmup_epoch, increment_epoch)
    return 2.0 if done else 1.0 + frac


def lr_lambda_factory(num_warmup_steps, num_training_steps, num_cycles=0.5, steps_sparsify=462, config=None):
    """The multiplier function of the reference's LambdaLR (:97-105)."""
    hold = bool(config is not None and config["only_lunif_epochs"] > 0)

    def lr_lambda(current_step):
        if current_step < steps_sparsify and hold:
            return 1.0          # constant LR while the warm-up ("sparsification") phase runs
        if current_step < num_warmup_steps:
            return float(current_step) / float(max(1, num_warmup_steps))
        progress = float(current_step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))

    return lr_lambda


def get_cosine_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, num_cycles=0.5, last_epoch=-1,
                                    steps_sparsify=462, config=None):
    """Cosine schedule with linear warm-up, same signature as the reference (:68-107).  Works with torch optimisers
    and with this package's AdamW (which exposes param_groups)."""
    return LambdaLR(optimizer, lr_lambda_factory(num_warmup_steps, num_training_steps, num_cycles, steps_sparsify, config), last_epoch)
