"""Optimizer tail of the reference training step, restated for the native modules
(/root/reference/scripts/pretrain_virtex.py:157-162 and virtex/factories.py:503-545):

    clip_grad_norm_(model.parameters(), 10.0)  ->  SGD(momentum .9, per-tensor lr / weight decay)
    ->  Lookahead(k = 5, alpha = .5)  ->  LinearWarmupCosineAnnealingLR.

Parameter grouping follows the reference exactly: one group per named parameter, weight decay 0
for names matching ``.*textual.(embedding|transformer).*(norm.*|bias)``, CNN_LR for names
containing ``cnn``.  Round 1 drives stock ``torch.optim.SGD(foreach=True)`` for the elementwise
update (the fused multi-tensor HIP kernel is the next row);
everything is device-side, no host synchronisation happens inside ``step()``.
"""
import math
import re
from typing import Iterable, List, Tuple

import torch

NO_DECAY = r".*textual.(embedding|transformer).*(norm.*|bias)"


def param_groups(named_parameters: Iterable[Tuple[str, torch.nn.Parameter]], cnn_lr=0.2, lr=0.001,
                 weight_decay=1e-4, no_decay=NO_DECAY) -> List[dict]:
    groups = []
    for name, p in named_parameters:
        wd = 0.0 if re.match(no_decay, name) else weight_decay
        groups.append({"params": [p], "lr": cnn_lr if "cnn" in name else lr, "weight_decay": wd})
    return groups


def lr_multiplier(step: int, total_steps: int, warmup_steps: int) -> float:
    """Linear warm-up then cos^2 decay (virtex/optim/lr_scheduler.py:174-183)."""
    if step < warmup_steps:
        return step / float(max(1, warmup_steps))
    frac = (step - warmup_steps) / (total_steps - warmup_steps)
    return max(0.0, math.cos(frac * (math.pi / 2)) ** 2)


class PretrainOptimizer:
    """SGD + Lookahead + warm-up/cosine schedule + global-norm clipping in one object."""

    def __init__(self, model: torch.nn.Module, cnn_lr=0.2, lr=0.001, weight_decay=1e-4, momentum=0.9,
                 clip_norm=10.0, lookahead_k=5, lookahead_alpha=0.5, total_steps=500000, warmup_steps=10000,
                 start_step=0):
        self.model = model
        groups = param_groups(model.named_parameters(), cnn_lr, lr, weight_decay)
        self.base_lrs = [g["lr"] for g in groups]
        self.sgd = torch.optim.SGD(groups, momentum=momentum, foreach=True)
        self.params = [g["params"][0] for g in self.sgd.param_groups]
        self.clip_norm, self.k, self.alpha = clip_norm, lookahead_k, lookahead_alpha
        self.slow = [p.detach().clone() for p in self.params]
        self.kc, self.step_idx = 0, start_step
        self.total_steps, self.warmup_steps = total_steps, warmup_steps
        self._set_lr()

    def _set_lr(self):
        mult = lr_multiplier(self.step_idx, self.total_steps, self.warmup_steps)
        for g, base in zip(self.sgd.param_groups, self.base_lrs):
            g["lr"] = base * mult

    def zero_grad(self):
        self.sgd.zero_grad(set_to_none=False)

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0):
        """`grad_scale` multiplies every gradient first (1/world_size after a SUM all-reduce)."""
        grads = [p.grad for p in self.params]
        if grad_scale != 1.0:
            torch._foreach_mul_(grads, grad_scale)
        if self.clip_norm is not None:
            torch.nn.utils.clip_grad_norm_(self.params, self.clip_norm, foreach=True)
        self.sgd.step()
        self.kc += 1
        if self.kc >= self.k:
            self.kc = 0
            data = [p.data for p in self.params]
            torch._foreach_mul_(data, self.alpha)
            torch._foreach_add_(data, self.slow, alpha=1.0 - self.alpha)
            torch._foreach_copy_(self.slow, data)
        self.step_idx += 1
        self._set_lr()
