"""`scaledreamer-system`: the training step of the reference's LightningModule without Lightning
(threestudio/systems/base.py:27-303 BaseSystem / BaseLift3DSystem, threestudio/systems/scaledreamer.py:14-170
StableDreamer.training_step, threestudio/systems/utils.py:19-53 optimizer parsing).

One process per GPU; when torch.distributed is initialised the field-parameter gradients are mean
all-reduced once per optimizer step over RCCL (what Lightning's DDP does in the reference, launch.py:233-240)
— see scaledreamer_amd/dist.py.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Any, Dict, Optional

import torch
import torch.nn as nn

from . import dist as asd_dist
from .base import Updateable, get_device
from .config import C, ConfigDict, parse_structured
from .registry import find, register


def dot(x, y):
    return torch.sum(x * y, -1, keepdim=True)


def binary_cross_entropy(inp, target):
    return -(target * torch.log(inp) + (1 - target) * torch.log(1 - inp)).mean()


class _LossTailFn(torch.autograd.Function):
    """sum_j w_j term_j + the per-ray regularisers (sparsity, opaque, z_variance) as one launch forward, one backward (include/asd_hip.h
    asd_loss_tail_fwd / _bwd) instead of ~25 tensor ops issued at the moment the autograd pass starts and the host has nothing queued."""

    @staticmethod
    def forward(ctx, opacity, z_variance, lams, weights, *terms):
        import ctypes as C

        from . import _lib

        op = opacity.detach().contiguous()
        zv = None if z_variance is None else z_variance.detach().contiguous()
        ts = [t.detach().reshape(1).float() for t in terms]
        out = torch.empty(5, device=op.device, dtype=torch.float32)
        n = len(ts)
        tab = (C.c_void_p * max(n, 1))(*[t.data_ptr() for t in ts])
        w = (C.c_float * max(n, 1))(*[float(x) for x in weights])
        _lib.check(_lib.lib().asd_loss_tail_fwd(tab, w, _lib.i32(n), _lib.ptr(op), _lib.ptr(zv), C.c_int64(op.numel()), _lib.f32(lams[0]),
                                                _lib.f32(lams[1]), _lib.f32(lams[2]), _lib.ptr(out), _lib.stream()))
        ctx.save_for_backward(op, out)
        ctx.meta = (lams, list(weights), opacity.shape, None if z_variance is None else z_variance.shape,
                    [t.shape for t in terms], [t.dtype for t in terms])
        total, values = out[0], out[1:4]
        ctx.mark_non_differentiable(values)
        return total, values

    @staticmethod
    def backward(ctx, d_total, _d_values):
        import ctypes as C

        from . import _lib

        op, out = ctx.saved_tensors
        lams, weights, op_shape, zv_shape, t_shapes, t_dtypes = ctx.meta
        n = len(weights)
        up = d_total.detach().reshape(1).float().contiguous()
        d_op = torch.empty_like(op)
        d_zv = torch.empty_like(op) if zv_shape is not None and ctx.needs_input_grad[1] else None
        d_terms = torch.empty(max(n, 1), device=op.device, dtype=torch.float32)
        w = (C.c_float * max(n, 1))(*[float(x) for x in weights])
        _lib.check(_lib.lib().asd_loss_tail_bwd(_lib.ptr(up), w, _lib.i32(n), _lib.ptr(op), C.c_int64(op.numel()), _lib.f32(lams[0]),
                                                _lib.f32(lams[1]), _lib.f32(lams[2]), _lib.ptr(out), _lib.ptr(d_terms), _lib.ptr(d_op),
                                                _lib.ptr(d_zv), _lib.stream()))
        g_terms = [d_terms[j].reshape(t_shapes[j]).to(t_dtypes[j]) if ctx.needs_input_grad[4 + j] else None for j in range(n)]
        return (d_op.view(op_shape), None if d_zv is None else d_zv.view(zv_shape), None, None, *g_terms)


def getattr_recursive(m, attr):
    for name in attr.split("."):
        m = getattr(m, name)
    return m


def parse_optimizer(config, model) -> torch.optim.Optimizer:
    """systems/utils.py:25-53: param groups addressed by dotted attribute path on the system."""
    if "params" in config and config["params"] is not None:
        params = []
        for name, args in config["params"].items():
            module = getattr_recursive(model, name)
            ps = module.parameters() if isinstance(module, nn.Module) else module
            params.append({"params": ps, "name": name, **args})
    else:
        params = model.parameters()
    if config["name"] == "FusedAdam":
        raise NotImplementedError("optimizer FusedAdam needs apex (systems/utils.py:42-45); use Adam, which runs fused here")
    if config["name"] == "Adan":
        from .optimizers import Adan

        return Adan(params, **dict(config.get("args", {})))
    args = dict(config.get("args", {}))
    if config["name"] in ("Adam", "AdamW") and next(model.parameters()).is_cuda and not any(args.get(k) for k in ("amsgrad", "maximize", "capturable")):
        # same update rule as the reference's torch.optim call, executed by the library's own multi-tensor kernel: one launch for all
        # five parameter groups (asd_adamw_f32) instead of ~20 foreach kernels per group
        from .optimizers import AdamW

        for k in ("fused", "foreach"):
            args.pop(k, None)
        if config["name"] == "Adam":
            args.setdefault("weight_decay", 0.0)        # torch.optim.Adam's default
        return AdamW(params, adam_l2=config["name"] == "Adam", **args)
    return getattr(torch.optim, config["name"])(params, **args)


@register("scaledreamer-system")
class StableDreamer(nn.Module, Updateable):
    @dataclass
    class Config:
        loggers: dict = field(default_factory=dict)
        loss: dict = field(default_factory=dict)
        optimizer: dict = field(default_factory=dict)
        scheduler: Optional[dict] = None
        weights: Optional[str] = None
        weights_ignore_modules: Optional[list] = None
        cleanup_after_validation_step: bool = False
        cleanup_after_test_step: bool = False
        geometry_type: str = ""
        geometry: dict = field(default_factory=dict)
        geometry_convert_from: Optional[str] = None
        geometry_convert_inherit_texture: bool = False
        geometry_convert_override: dict = field(default_factory=dict)
        material_type: str = ""
        material: dict = field(default_factory=dict)
        background_type: str = ""
        background: dict = field(default_factory=dict)
        renderer_type: str = ""
        renderer: dict = field(default_factory=dict)
        guidance_type: str = ""
        guidance: dict = field(default_factory=dict)
        prompt_processor_type: str = ""
        prompt_processor: dict = field(default_factory=dict)
        exporter_type: str = "mesh-exporter"
        exporter: dict = field(default_factory=dict)
        stage: str = "coarse"
        visualize_samples: bool = False
        validation_via_video: bool = False

    cfg: Config

    def __init__(self, cfg, guidance_backend=None, prompt_utils=None) -> None:
        super().__init__()
        from . import plugins  # noqa: F401

        self.cfg = parse_structured(self.Config, cfg)
        self.device = get_device()
        self.true_global_step, self.current_epoch = 0, 0
        # BaseLift3DSystem.configure (systems/base.py:249-303)
        self.geometry = find(self.cfg.geometry_type)(self.cfg.geometry)
        self.material = find(self.cfg.material_type)(self.cfg.material)
        self.background = find(self.cfg.background_type)(self.cfg.background)
        self.renderer = find(self.cfg.renderer_type)(self.cfg.renderer, geometry=self.geometry, material=self.material,
                                                      background=self.background)
        self.to(self.device)
        # on_fit_start (scaledreamer.py:38-45): guidance + prompt utils are training-only, not nn.Modules
        self.guidance = find(self.cfg.guidance_type)(self.cfg.guidance, backend=guidance_backend) if self.cfg.guidance_type else None
        self.prompt_utils = prompt_utils
        self.optimizer = parse_optimizer(self.cfg.optimizer, self) if self.cfg.optimizer else None
        self.logged: Dict[str, Any] = {}

    def C(self, value: Any) -> float:
        return C(value, self.current_epoch, self.true_global_step)

    def log(self, name, value, **kw):
        self.logged[name] = value

    def forward(self, batch: Dict[str, Any]) -> Dict[str, Any]:
        return self.renderer(**batch)      # (a LazyOutputs: copying it with {**out} would force its deferred entries)

    def on_train_batch_start(self, batch_idx: int = 0):
        """systems/base.py:180-184: per-step update hooks (occupancy grid, timestep annealing, resolution).  The guidance is a
        plain attribute of an Updateable, so the recursion of do_update_step already reaches it."""
        self.do_update_step(self.current_epoch, self.true_global_step)

    def on_train_batch_end(self, batch_idx: int = 0):
        """systems/base.py:120-125"""
        self.do_update_step_end(self.current_epoch, self.true_global_step)

    # ---- loss assembly (scaledreamer.py:62-126; multiprompt_radience_field_generator.py:142-212) ----------------------------------
    # Every regulariser is an entry of REGULARISERS: name -> (needs, fn(out) -> scalar).  A term is evaluated iff its
    # `lambda_<name>` is configured and positive at the current step (the lambdas are C() schedules); `needs` is the output key
    # whose absence is the reference's ValueError.  `lambda_eikonal` is optional in the config (hasattr test in the reference),
    # the other four are read unconditionally there, so a config without them is a KeyError here as well.
    GEOMETRY_PASS_WEIGHT = 0.5       # scaledreamer.py:122 ("hard-coded lambda"); 0.2 in the multi-prompt system (:203)
    REGULARISERS = {
        "orient": ("normal", "Normal is required for orientation loss, no normal is found in the output.",
                   lambda o: (o["weights"].detach() * dot(o["normal"], o["t_dirs"]).clamp_min(0.0) ** 2).sum() / (o["opacity"] > 0).sum()),
        "sparsity": ("opacity", "opacity is required for the sparsity loss.",
                     lambda o: (o["opacity"] ** 2 + 0.01).sqrt().mean()),
        "opaque": ("opacity", "opacity is required for the opaque loss.",
                   lambda o: (lambda oc: binary_cross_entropy(oc, oc))(o["opacity"].clamp(1.0e-3, 1.0 - 1.0e-3))),
        "z_variance": ("z_variance", "z_variance is required for z_variance loss, no z_variance is found in the output.",
                       lambda o: o["z_variance"][o["opacity"] > 0.5].mean()),
        "eikonal": ("sdf_grad", "sdf is required for eikonal loss, no sdf is found in the output.",
                    lambda o: ((torch.linalg.norm(o["sdf_grad"], ord=2, dim=-1) - 1.0) ** 2).mean()),
    }
    OPTIONAL_LAMBDAS = ("eikonal",)

    FUSED_REGULARISERS = ("sparsity", "opaque", "z_variance")     # the per-ray terms of _LossTailFn, in its lambda order

    def _guidance_terms(self, image, batch, prefix: str, weight: float, rgb_as_latents: bool = False):
        """[(value, weight * lambda)] of the guidance's loss_* entries"""
        terms = []
        for name, value in self.guidance(image, self.prompt_utils, **batch, rgb_as_latents=rgb_as_latents).items():
            self.log(f"train/{prefix}{name}", value)
            if name.startswith("loss_"):
                terms.append((value, weight * self.C(self.cfg.loss["lambda_" + name[len("loss_"):]])))
        return terms

    def _regulariser_terms(self, out, fused: bool):
        """[(value, lambda)] of the regularisers evaluated as tensor ops, and the lambdas of the ones left to _LossTailFn (fused)"""
        terms, lams = [], {}
        for name, (needs, message, fn) in self.REGULARISERS.items():
            key = "lambda_" + name
            if name in self.OPTIONAL_LAMBDAS and key not in self.cfg.loss:
                continue
            lam = self.C(self.cfg.loss[key])
            if not lam > 0:
                continue
            if needs not in out:
                raise ValueError(message)
            if fused and name in self.FUSED_REGULARISERS:
                lams[name] = lam
                continue
            value = fn(out)
            self.log(f"train/loss_{name}", value)
            terms.append((value, lam))
            if name == "eikonal":
                self.log("train/inv_std", out["inv_std"])
        return terms, lams

    def _assemble_loss(self, terms, out, lams, fused: bool):
        """sum of weight * value over `terms` plus the fused per-ray regularisers `lams`; in the reference's order of additions (tensor
        ops) when `fused` is off — CPU tensors, ASD_LOSS_TAIL=0, no fp32 device opacity, more terms than one launch takes"""
        ASD_LOSS_MAX_TERMS = 8                     # csrc/asd_glue.hip: the term table of one asd_loss_tail launch

        opacity = out.get("opacity") if fused else None
        if (not fused or not torch.is_tensor(opacity) or not opacity.is_cuda or opacity.dtype != torch.float32 or len(terms) > ASD_LOSS_MAX_TERMS
                or not all(torch.is_tensor(v) and v.is_cuda for v, _ in terms)):
            terms = list(terms)
            for name, lam in lams.items():          # regularisers that were left to the fused tail: as tensor ops after all
                value = self.REGULARISERS[name][2](out)
                self.log(f"train/loss_{name}", value)
                terms.append((value, lam))
            total = 0.0
            for value, w in terms:
                total = total + value * w
            return total
        z_var = out["z_variance"] if "z_variance" in lams else None
        lam3 = tuple(float(lams.get(k, 0.0)) for k in self.FUSED_REGULARISERS)
        total, values = _LossTailFn.apply(opacity, z_var, lam3, tuple(float(w) for _, w in terms), *[v for v, _ in terms])
        for i, k in enumerate(self.FUSED_REGULARISERS):
            if k in lams:
                self.log(f"train/loss_{k}", values[i])
        return total

    def _rgb_as_latents(self) -> bool:
        return False

    def training_step(self, batch, batch_idx: int = 0):
        stage = self.cfg.stage
        if stage not in ("coarse", "coarse+geometry"):
            # 'geometry' / 'texture' (mesh stages: normal consistency, laplacian) are not on the ASD hot path
            raise ValueError(f"stage {stage!r}: only the NeRF stages 'coarse' and 'coarse+geometry' are implemented")
        out = self(batch)
        terms = self._guidance_terms(out["comp_rgb"], batch, "", 1.0, self._rgb_as_latents())
        fused = ("opacity" in out and torch.is_tensor(out["opacity"]) and out["opacity"].is_cuda and out["opacity"].dtype == torch.float32
                 and os.environ.get("ASD_LOSS_TAIL", "1") != "0")        # =0: the tensor-op form (A/B, tools/r5_ab_env.sh)
        reg_terms, lams = self._regulariser_terms(out, fused)
        terms = terms + reg_terms
        if stage == "coarse+geometry":   # second guidance pass on the normal image
            normal_img = torch.nan_to_num(out["comp_normal"], nan=0.0, posinf=0.0, neginf=0.0)
            terms = terms + self._guidance_terms(normal_img, batch, "shape_", self.GEOMETRY_PASS_WEIGHT)
        return {"loss": self._assemble_loss(terms, out, lams, fused)}

    def gradient_exchange(self):
        """the DP exchange object of this system (None on a single process): created on first use, after the process group."""
        if not asd_dist.is_distributed():
            return None
        if getattr(self, "_exchange", None) is None:
            self._exchange = asd_dist.GradientExchange([p for grp in self.optimizer.param_groups for p in grp["params"]])
        return self._exchange

    # trainer.accumulate_grad_batches of the reference's YAMLs (Lightning; asd_mv_triplane_transformer_10k.yaml:129: 2 on 8 GPUs,
    # _1GPU.yaml:131: 8): the optimizer steps every k-th batch on the SUM of the k gradients of loss / k, the DDP exchange runs in the k-th
    # backward only (no_sync before), and global_step — what every C() schedule and update_step sees — counts optimizer steps.
    accumulate_grad_batches: int = 1

    def train_one_step(self, batch) -> torch.Tensor:
        """on_train_batch_start -> training_step -> backward (gradient all-reduces launched from its hooks) -> optimizer.step; with
        accumulate_grad_batches = k the exchange and the optimizer run on every k-th call."""
        k = max(1, int(self.accumulate_grad_batches))
        micro = getattr(self, "_micro_batch", 0)
        first, last = micro == 0, micro == k - 1
        self.on_train_batch_start()
        ex = self.gradient_exchange()
        if ex is not None:
            ex.prepare(zero=first, sync=last)
        elif first:
            self.optimizer.zero_grad(set_to_none=True)
        loss = self.training_step(batch)["loss"]
        (loss if k == 1 else loss / k).backward()
        if last:
            if ex is not None:
                ex.finish()
            self.optimizer.step()
        self.on_train_batch_end()
        if last:
            self.true_global_step += 1
        self._micro_batch = 0 if last else micro + 1
        return loss.detach()
