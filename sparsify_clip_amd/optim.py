"""Fused AdamW over the model's flat parameter buffer (replaces torch.optim.AdamW of reference sparsify_clip.py:730).

torch defaults, as the reference uses them: betas (0.9, 0.999), eps 1e-8, weight_decay 0.01 on EVERY tensor
(LayerNorm, biases and the learnable temperature included).  One kernel launch updates all 151 M parameters and
writes the bf16 shadow the GEMMs read; the [in,out] weight copies are rebuilt right after.  Subclasses
torch.optim.Optimizer only so that LambdaLR (the reference's scheduler) can drive `param_groups[...]["lr"]`.
"""
from __future__ import annotations

import torch

from . import ops


class AdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, extra_params=()):
        self.model = model
        self.trainable = model.flat[: model.n_trainable]
        extra = [p for p in extra_params]
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        groups = [{"params": [self.trainable]}]
        if extra:
            groups.append({"params": extra})
        super().__init__(groups, defaults)
        self.m = torch.zeros_like(self.trainable)
        self.v = torch.zeros_like(self.trainable)
        self.extra_state = {id(p): (torch.zeros(1, device=model.device), torch.zeros(1, device=model.device), torch.zeros(1, device=model.device),
                                    torch.zeros(1, device=model.device)) for p in extra}   # (value on device, grad, m, v)
        self.steps = 0
        self.extra_steps = {id(p): 0 for p in extra}

    def zero_grad(self, set_to_none=True):
        self.model.zero_grad()
        for g in self.param_groups[1:]:
            for p in g["params"]:
                p.grad = None

    @torch.no_grad()
    def step(self, closure=None):
        g0 = self.param_groups[0]
        self.steps += 1
        m = self.model
        shadow = m.flat_bf16[: m.n_trainable] if m.flat_bf16 is not None else None
        m.wait_wt()   # the previous step's overlapped transposes read the masters this step overwrites
        ops.adamw_step(self.trainable, m.flat_grad[: m.n_trainable], self.m, self.v, shadow, g0["lr"], g0["betas"][0], g0["betas"][1],
                       g0["eps"], g0["weight_decay"], self.steps)
        m.refresh_shadows(full=False, overlap=True)
        # scalar extras (the learnable temperature): same kernel on a 1-element buffer; skipped when no gradient arrived,
        # exactly as torch's AdamW skips parameters whose .grad is None
        for g in self.param_groups[1:]:
            for p in g["params"]:
                if p.grad is None:
                    continue
                val, grad, em, ev = self.extra_state[id(p)]
                self.extra_steps[id(p)] += 1
                val.copy_(p.detach().reshape(1))
                grad.copy_(p.grad.detach().reshape(1))
                ops.adamw_step(val, grad, em, ev, None, g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self.extra_steps[id(p)])
                p.data.copy_(val.reshape(p.shape))
