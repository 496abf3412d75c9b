"""Adan (Xie et al. 2022, arXiv:2208.06677) with the update rule of threestudio/systems/optimizers.py:23-315, used by the
triplane-transformer config (asd_mv_triplane_transformer_10k.yaml:101-107).  Per parameter, with g the (globally clipped)
gradient and d = g - g_prev (0 on a group's first step):
    m <- b1 m + (1-b1) g         v <- b2 v + (1-b2) d         n <- b3 n + (1-b3) (g + b2 d)^2
    p <- (p - lr/(1-b1^t) * m/den - lr b2/(1-b2^t) * v/den) / (1 + lr wd),   den = sqrt(n)/sqrt(1-b3^t) + eps
(no_prox: p is multiplied by (1 - lr wd) before the step instead).  Written with torch._foreach ops over a group.
"""
from __future__ import annotations

import math

import torch
from torch.optim import Optimizer


class Adan(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.98, 0.92, 0.99), eps=1e-8, weight_decay=0.0, max_grad_norm=0.0, no_prox=False,
                 foreach: bool = True):
        if max_grad_norm < 0.0:
            raise ValueError(f"Invalid Max grad norm: {max_grad_norm}")
        if lr < 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")
        if eps < 0.0:
            raise ValueError(f"Invalid epsilon value: {eps}")
        for i, b in enumerate(betas):
            if not 0.0 <= b < 1.0:
                raise ValueError(f"Invalid beta parameter at index {i}: {b}")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, max_grad_norm=max_grad_norm,
                                      no_prox=no_prox, foreach=foreach))

    @torch.no_grad()
    def restart_opt(self):
        for group in self.param_groups:
            group["step"] = 0
            for p in group["params"]:
                if p.requires_grad:
                    st = self.state[p]
                    st["exp_avg"], st["exp_avg_sq"], st["exp_avg_diff"] = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        clip = 1.0
        if self.defaults["max_grad_norm"] > 0:
            sq = [p.grad.pow(2).sum() for g in self.param_groups for p in g["params"] if p.grad is not None]
            norm = torch.sqrt(torch.stack(sq).sum())
            clip = float(torch.clamp(self.defaults["max_grad_norm"] / (norm + self.param_groups[-1]["eps"]), max=1.0))
        for group in self.param_groups:
            b1, b2, b3 = group["betas"]
            group["step"] = group.get("step", 0) + 1
            t = group["step"]
            ps, gs, ms, ns, vs, prevs = [], [], [], [], [], []
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["exp_avg"], st["exp_avg_sq"], st["exp_avg_diff"] = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
                if "neg_pre_grad" not in st or t == 1:   # state key kept for checkpoint compatibility: -(previous clipped gradient)
                    st["neg_pre_grad"] = p.grad.clone().mul_(-clip)
                ps.append(p); gs.append(p.grad); ms.append(st["exp_avg"]); ns.append(st["exp_avg_sq"]); vs.append(st["exp_avg_diff"])
                prevs.append(st["neg_pre_grad"])
            if not ps:
                continue
            lr, wd = group["lr"], group["weight_decay"]
            torch._foreach_mul_(gs, clip)
            torch._foreach_add_(prevs, gs)                                  # d = g - g_prev
            torch._foreach_mul_(ms, b1); torch._foreach_add_(ms, gs, alpha=1 - b1)
            torch._foreach_mul_(vs, b2); torch._foreach_add_(vs, prevs, alpha=1 - b2)
            torch._foreach_mul_(prevs, b2); torch._foreach_add_(prevs, gs)   # g + b2 d
            torch._foreach_mul_(ns, b3); torch._foreach_addcmul_(ns, prevs, prevs, value=1 - b3)
            den = torch._foreach_sqrt(ns)
            torch._foreach_div_(den, math.sqrt(1.0 - b3 ** t))
            torch._foreach_add_(den, group["eps"])
            if group["no_prox"]:
                torch._foreach_mul_(ps, 1 - lr * wd)
            torch._foreach_addcdiv_(ps, ms, den, value=-lr / (1.0 - b1 ** t))
            torch._foreach_addcdiv_(ps, vs, den, value=-lr * b2 / (1.0 - b2 ** t))
            if not group["no_prox"]:
                torch._foreach_div_(ps, 1 + lr * wd)
            torch._foreach_zero_(prevs)
            torch._foreach_add_(prevs, gs, alpha=-1.0)
        return loss
