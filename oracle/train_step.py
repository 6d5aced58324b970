"""Oracle: the reference's training step on CPU in fp32 (test infrastructure, see oracle/__init__.py).

Restates the body of the reference's step loop (sparsify_clip.py:753-969) with plain torch on the oracle model:
forward both encoders, L2-normalise (:772-773), loss_type dispatch (:778-938), zero_grad, backward, AdamW with
torch defaults over all parameters (+ the learnable temperature, :727-730), LambdaLR step (:969).  Used by the
step-parity tests and as the `cpu_baseline` of bench.py.
"""
from __future__ import annotations

import torch
from torch.optim.lr_scheduler import LambdaLR

from . import dispatch as D
from .clip_model import create_model
from .loss_head import normalize_rows
from .schedules import lr_multiplier


class CpuTrainer:
    def __init__(self, config, steps_per_epoch, model=None):
        self.config = config
        self.model = model or create_model(config["model"], seed=config["seed"])
        self.model.train()
        self.temperature = config["anchor_temperature"]
        params = [p for n, p in self.model.named_parameters()]
        if config["anchor_temperature_learnable"]:
            self.temperature = torch.nn.Parameter(torch.tensor(self.temperature, dtype=torch.float32))
            params.append(self.temperature)
        self.optimizer = torch.optim.AdamW(params, lr=config["learning_rate"])
        self.t_total = steps_per_epoch * config["epochs"]
        warm = int(0.20 * self.t_total)
        self.scheduler = LambdaLR(self.optimizer, lambda s: lr_multiplier(s, warm, self.t_total, config["only_lunif_epochs"]))
        self.current_batch, self.epoch = 0, 0

    def step(self, images, tokens):
        self.current_batch += 1
        img = normalize_rows(self.model.encode_image(images))
        txt = normalize_rows(self.model.encode_text(tokens))
        loss, beta, alpha = D.compose_loss(self.config, img, txt, self.temperature, self.epoch, self.current_batch, self.t_total)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        self.scheduler.step()
        return loss.detach()
