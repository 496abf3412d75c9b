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


import numpy as np

# include/asd_hip.h asd_opt_tensor, field for field (C alignment: 80 bytes)
_OPT_DTYPE = np.dtype([("p", "u8"), ("g", "u8"), ("m", "u8"), ("v", "u8"), ("n2", "u8"), ("prev", "u8"), ("n", "i8"), ("lr", "f4"),
                       ("weight_decay", "f4"), ("bias_correction1", "f4"), ("bias_correction2", "f4"), ("bias_correction2_sqrt", "f4")], align=True)


def _check_f32(name: str, t: torch.Tensor) -> None:
    if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
        raise TypeError(f"fused optimizer: {name} must be a contiguous fp32 device tensor")


class _Table:
    """The tensor table of a fused optimizer launch (asd_opt_tensor[]) for ONE set of parameters: the parameter / state pointers and sizes
    are written once, a step only fills the gradient pointers and the per-step scalars (numpy columns) — the per-tensor Python of building
    the table from scratch was 10 us per parameter, 1.3 ms of idle GPU per step on the 130 parameters of the StyleGAN-3D generator.  The
    library copies the descriptors into kernel arguments at launch, so the host buffer is free for the next step on return."""

    def __init__(self, fn_name: str, params, **state_cols):
        from ._lib import OptTensor

        assert C.sizeof(OptTensor) == _OPT_DTYPE.itemsize
        self.fn_name, self.params = fn_name, list(params)
        self.keep = [list(col) for col in state_cols.values()]       # the state tensors stay alive with the table
        self.arr = np.zeros(len(self.params), _OPT_DTYPE)
        for p in self.params:
            _check_f32("p", p.data)
        self.arr["p"] = [p.data_ptr() for p in self.params]
        self.arr["n"] = [p.numel() for p in self.params]
        for k, col in state_cols.items():
            for t in col:
                _check_f32(k, t)
            self.arr[k] = [t.data_ptr() for t in col]

    def same(self, params, *state_cols) -> bool:
        """the table still describes these parameters and these state tensors (identity, not value)"""
        if len(params) != len(self.params) or any(a is not b for a, b in zip(params, self.params)):
            return False
        # a Parameter whose storage was replaced (`p.data = ...`, `module.to()`, `set_`) keeps its identity: compare the addresses too
        if any(p.data_ptr() != int(q) or p.numel() != int(n) for p, q, n in zip(params, self.arr["p"], self.arr["n"])):
            return False
        return all(len(c) == len(k) and all(a is b for a, b in zip(c, k)) for c, k in zip(state_cols, self.keep))

    def launch(self, grads, lr, wd, bc1, bc2, bc2s, *scalars) -> None:
        from ._lib import check, lib, stream

        for g in grads:
            if not (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous()):
                raise TypeError("fused optimizer: g must be a contiguous fp32 device tensor")
        a = self.arr
        a["g"] = [g.data_ptr() for g in grads]
        a["lr"], a["weight_decay"], a["bias_correction1"], a["bias_correction2"], a["bias_correction2_sqrt"] = lr, wd, bc1, bc2, bc2s
        check(getattr(lib(), self.fn_name)(C.c_void_p(a.ctypes.data), C.c_int32(len(a)), *scalars, stream()))
        # the kernel wrote the parameters through raw pointers: tell autograd (saved-tensor checks) and every cache keyed on
        # `p._version` (the tri-plane transformer's packed operand planes, generators.TriplaneTransformer._tritx_state)
        torch.autograd.graph.increment_version(self.params)


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
        self.__dict__["_tables"] = {}          # the cached tensor tables point into the old state
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
        # parameters with a gradient, per distinct (betas, eps, rule): normally one set, the same one every step
        sets = {}
        for gi, group in enumerate(self.param_groups):
            key = (*group["betas"], group["eps"], bool(group.get("adam_l2", self.defaults["adam_l2"])))
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("AdamW does not support sparse gradients")
                ps, gis = sets.setdefault(key, ([], []))
                ps.append(p)
                gis.append(gi)
        tables = self.__dict__.setdefault("_tables", {})
        for key, (ps, gis) in sets.items():
            b1, b2, eps, l2 = key
            tab = tables.get(key)
            if tab is not None:       # still these parameters, these state tensors, and a `step` nobody rewrote behind the mirror
                cur = [self.state[p] for p in ps] if all(p in self.state for p in ps) else None
                if (cur is None or not tab[0].same(ps, [st.get("exp_avg") for st in cur], [st.get("exp_avg_sq") for st in cur])
                        or any(st.get("step") is not s0 for st, s0 in zip(cur, tab[1])) or float(tab[1][0]) != tab[2][0]):
                    tab = None
            if tab is None:
                for p in ps:
                    st = self.state[p]
                    if not st:
                        st["step"] = torch.tensor(0.0)
                        st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format), torch.zeros_like(p, memory_format=torch.preserve_format)
                    if not torch.is_tensor(st["step"]):          # old torch state dicts hold a python int
                        st["step"] = torch.tensor(float(st["step"]))
                    elif st["step"].is_cuda:                      # Optimizer.load_state_dict moves per-parameter state next to the parameter:
                        st["step"] = st["step"].cpu()             # one read at load time, not one per step
                steps = [self.state[p]["step"] for p in ps]
                tab = (_Table("asd_adamw_f32", ps, m=[self.state[p]["exp_avg"] for p in ps], v=[self.state[p]["exp_avg_sq"] for p in ps]),
                       steps, np.array([float(t) for t in steps], dtype=np.float64))
                tables[key] = tab
            table, steps, t = tab
            torch._foreach_add_(steps, 1.0)                       # the state's `step` tensors (host), one call for all of them
            t += 1.0                                              # ... and their mirror
            lr = np.array([self.param_groups[gi]["lr"] for gi in gis], dtype=np.float64)
            wd = np.array([self.param_groups[gi]["weight_decay"] for gi in gis], dtype=np.float64)
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
            table.launch(grads, lr, wd, 1.0 - b1 ** t, 1.0, np.sqrt(1.0 - b2 ** t), C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_int32(int(l2)))
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
            tables = self.__dict__.setdefault("_tables", {})
            tab = tables.get(id(group))
            if tab is None or not tab.same(ps, ms, vs, ns, prevs):
                tab = tables[id(group)] = _Table("asd_adan_f32", ps, m=ms, v=vs, n2=ns, prev=prevs)
            gs = [g if g.is_contiguous() else g.contiguous() for g in gs]
            tab.launch(gs, lr, wd, 1.0 - b1 ** t, 1.0 - b2 ** t, math.sqrt(1.0 - b3 ** t), C.c_float(b1), C.c_float(b2), C.c_float(b3),
                       C.c_float(group["eps"]), C.c_float(clip), C.c_int32(int(group["no_prox"])))
        return loss
