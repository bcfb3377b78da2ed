"""Oracle: optimizer steps and LR schedules.  TEST INFRASTRUCTURE ONLY.

Reference call sites: train_hdf5_ddp.py:213-218 (Adam / AdamW / apex FusedLAMB), :358,364 (zero_grad, step),
:245-260 + utils/parsing_helpers.py:27-37 (MultiStepLR, optional GradualWarmupScheduler), :369-371 (stepping).

Adam/AdamW restate torch.optim's single-tensor algorithm and are PINNED against golden
vectors produced with torch.optim itself through the reference model (tests/golden).
LAMB and warm-up are PARITY UNPINNED: apex and pytorch-gradual-warmup-lr are neither vendored
in the reference nor installable here.  Their definition below is this project's own, taken from
You et al. 2019 and apex's documented FusedLAMB defaults (bias correction on, max_grad_norm 1.0,
adam_w_mode on, betas (0.9, 0.999), use_nvlamb off: trust ratio only where weight_decay != 0).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import torch


class OracleOptimizer:
    def __init__(self, params: Sequence[torch.Tensor], kind: str, lr: float, eps: float = 1e-8,
                 weight_decay: float = 0.0, betas=(0.9, 0.999), max_grad_norm: float = 1.0):
        if kind not in ("Adam", "AdamW", "LAMB"):
            raise NotImplementedError("Error, optimizer {} not supported".format(kind))   # train_hdf5_ddp.py:220
        self.params = list(params)
        self.kind, self.lr, self.eps, self.wd, self.betas = kind, lr, eps, weight_decay, betas
        self.max_grad_norm = max_grad_norm
        self.step_count = 0
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]

    @torch.no_grad()
    def step(self, grads: Sequence[torch.Tensor]) -> None:
        self.step_count += 1
        b1, b2 = self.betas
        t = self.step_count
        bc1 = 1.0 - b1 ** t
        bc2 = 1.0 - b2 ** t
        if self.kind == "LAMB":
            gnorm = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads))
            clip = gnorm / self.max_grad_norm if gnorm > self.max_grad_norm else 1.0
        for p, g, m, v in zip(self.params, grads, self.m, self.v):
            if self.kind == "Adam":
                # torch.optim.Adam: L2 penalty folded into the gradient
                g = g.add(p, alpha=self.wd) if self.wd != 0 else g
            elif self.kind == "AdamW":
                p.mul_(1.0 - self.lr * self.wd)
            else:
                g = g / clip
            m.lerp_(g, 1.0 - b1)
            v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
            if self.kind in ("Adam", "AdamW"):
                denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
                p.addcdiv_(m, denom, value=-self.lr / bc1)
            else:
                upd = (m / bc1) / ((v / bc2).sqrt() + self.eps) + self.wd * p
                wn = float(p.double().norm())
                un = float(upd.double().norm())
                # apex multi_tensor_lamb stage 2 with use_nvlamb=False (FusedLAMB's default): the trust ratio is applied only to
                # tensors whose weight decay is non-zero
                ratio = wn / un if (self.wd != 0 and wn > 0 and un > 0) else 1.0
                p.add_(upd, alpha=-self.lr * ratio)


class MultiStepSchedule:
    """torch.optim.lr_scheduler.MultiStepLR as the reference drives it (parsing_helpers.py:27-35,
    train_hdf5_ddp.py:246,369-371), restated with explicit state.

    The scheduler is *recursive*: it multiplies whatever LR the optimizer currently holds by gamma each
    time its counter lands on a milestone.  Its constructor already performs one step, so with
    ``last_step = s`` the counter starts at s+1 and the LR the loop reads at its i-th iteration
    (get_last_lr() BEFORE scheduler.step(), :370-371) belongs to counter s+1+i.  A fresh run therefore
    uses the decayed LR from the m-th optimizer step on for a milestone m.  On resume the optimizer's own
    (already decayed) LR comes from the checkpoint; ``current_lr`` is that value.
    """

    def __init__(self, current_lr: float, milestones: Sequence[int], gamma: float, last_step: int = 0):
        self.milestones = list(milestones)
        self.gamma = gamma
        self.lr = current_lr
        self.counter = last_step
        self.step()                      # _initial_step()

    def get_last_lr(self) -> float:
        return self.lr

    def step(self) -> None:
        self.counter += 1
        n = self.milestones.count(self.counter)
        if n:
            self.lr = self.lr * self.gamma ** n


def warmup_lr(start_lr: float, factor: float, warmup_steps: int, step: int, after_lr: float) -> float:
    """PARITY UNPINNED.  Linear ramp start_lr -> start_lr*factor over warmup_steps (0 -> start_lr when factor == 1.0, as
    pytorch-gradual-warmup-lr's get_lr does), then the wrapped schedule * factor."""
    if warmup_steps > 0 and step <= warmup_steps:
        if factor == 1.0:
            return start_lr * (float(step) / warmup_steps)
        return start_lr * (1.0 + (factor - 1.0) * step / warmup_steps)
    return after_lr * factor


def parse_lr_schedule(arg: Dict[str, str]):
    """parsing_helpers.py:31-37."""
    if arg["type"] == "multistep":
        return [int(x) for x in arg["milestones"].split()], float(arg["decay_rate"])
    raise ValueError("Error, scheduler type {} not supported.".format(arg["type"]))
