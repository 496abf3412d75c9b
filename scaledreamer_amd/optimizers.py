"""Adan (Xie et al. 2022, arXiv:2208.06677) with the update rule of threestudio/systems/optimizers.py:23-315, used by the
triplane-transformer config (asd_mv_triplane_transformer_10k.yaml:101-107).  Per parameter, with g the (globally clipped)
gradient and d = g - g_prev (0 on a group's first step):
    m <- b1 m + (1-b1) g         v <- b2 v + (1-b2) d         n <- b3 n + (1-b3) (g + b2 d)^2
    p <- (p - lr/(1-b1^t) * m/den - lr b2/(1-b2^t) * v/den) / (1 + lr wd),   den = sqrt(n)/sqrt(1-b3^t) + eps
(no_prox: p is multiplied by (1 - lr wd) before the step instead).

Both optimizers of the reference's configs run as multi-tensor fused HIP kernels (csrc/optim.hip: asd_adamw_f32, asd_adan_f32):
one launch covers every parameter of every group.  `AdamW` here is torch.optim.AdamW / Adam with the same state layout
(`step`, `exp_avg`, `exp_avg_sq`), so optimizer state dicts move between the two.
"""
from __future__ import annotations

import ctypes as C
import math

import torch
from torch.optim import Optimizer


def _launch(fn_name: str, entries, *scalars) -> None:
    """entries: list of dicts with p, g, m, v[, n2, prev], lr, wd, bc1, bc2, bc2s"""
    from ._lib import OptTensor, check, lib, stream

    arr = (OptTensor * len(entries))()
    for a, e in zip(arr, entries):
        for k in ("p", "g", "m", "v", "n2", "prev"):
            t = e.get(k)
            if t is not None:
                if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
                    raise TypeError(f"fused optimizer: {k} must be a contiguous fp32 device tensor")
                setattr(a, k, t.data_ptr())
        a.n, a.lr, a.weight_decay = e["p"].numel(), e["lr"], e["wd"]
        a.bias_correction1, a.bias_correction2, a.bias_correction2_sqrt = e["bc1"], e.get("bc2", 1.0), e["bc2s"]
    check(getattr(lib(), fn_name)(arr, C.c_int32(len(entries)), *scalars, stream()))


class AdamW(Optimizer):
    """torch.optim.AdamW (decoupled weight decay) or, with adam_l2=True, torch.optim.Adam — amsgrad / maximize / capturable unsupported
    (the reference never sets them).  The whole step is ONE kernel launch per 24 tensors."""

    _TORCH_ONLY = ("amsgrad", "maximize", "capturable", "differentiable", "foreach", "fused")     # torch.optim.AdamW keywords

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, adam_l2: bool = False, **torch_kw):
        unknown = sorted(k for k in torch_kw if k not in self._TORCH_ONLY)
        if unknown:
            raise TypeError(f"AdamW.__init__() got unexpected keyword arguments {unknown}")
        bad = sorted(k for k in ("amsgrad", "maximize", "capturable", "differentiable") if torch_kw.get(k))
        if bad:
            raise NotImplementedError(f"fused AdamW: unsupported options {bad}")
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError("Invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, adam_l2=adam_l2))

    def __setstate__(self, state):
        """load_state_dict replaces param_groups with the saved ones: a state dict written by torch.optim.AdamW / Adam (or a reference
        checkpoint) has no `adam_l2` key — keep this optimizer's own rule — and may carry options this kernel does not implement."""
        super().__setstate__(state)
        for group in self.param_groups:
            group.setdefault("adam_l2", self.defaults["adam_l2"])
            bad = sorted(k for k in ("amsgrad", "maximize", "capturable", "differentiable") if group.get(k))
            if bad:
                raise NotImplementedError(f"fused AdamW: the loaded state uses unsupported options {bad}")

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        by_hyper = {}
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("AdamW does not support sparse gradients")
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format), torch.zeros_like(p, memory_format=torch.preserve_format)
                if not torch.is_tensor(st["step"]):          # old torch state dicts hold a python int
                    st["step"] = torch.tensor(float(st["step"]))
                elif st["step"].is_cuda:                      # Optimizer.load_state_dict moves per-parameter state next to the parameter:
                    st["step"] = st["step"].cpu()             # one read at load time, not one per step
                st["step"] += 1
                t = int(st["step"])
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                by_hyper.setdefault((b1, b2, group["eps"], bool(group.get("adam_l2", self.defaults["adam_l2"]))), []).append(
                    dict(p=p.data, g=g, m=st["exp_avg"], v=st["exp_avg_sq"], lr=group["lr"], wd=group["weight_decay"],
                         bc1=1.0 - b1 ** t, bc2s=math.sqrt(1.0 - b2 ** t)))
        for (b1, b2, eps, l2), entries in by_hyper.items():
            _launch("asd_adamw_f32", entries, C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_int32(int(l2)))
        return loss


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
            # one fused launch per 24 tensors (csrc/optim.hip: adan_kernel); device tensors only — the torch restatement of this
            # update lives in oracle/adan_ref.py (test infrastructure, pinned by tests/golden/adan_steps.npz)
            entries = [dict(p=p.data, g=g, m=m, v=v, n2=n, prev=pr, lr=lr, wd=wd, bc1=1.0 - b1 ** t, bc2=1.0 - b2 ** t, bc2s=math.sqrt(1.0 - b3 ** t))
                       for p, g, m, n, v, pr in zip(ps, gs, ms, ns, vs, prevs)]
            _launch("asd_adan_f32", entries, C.c_float(b1), C.c_float(b2), C.c_float(b3), C.c_float(group["eps"]), C.c_float(clip),
                    C.c_int32(int(group["no_prox"])))
        return loss
