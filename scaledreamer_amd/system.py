"""`scaledreamer-system`: the training step of the reference's LightningModule without Lightning
(threestudio/systems/base.py:27-303 BaseSystem / BaseLift3DSystem, threestudio/systems/scaledreamer.py:14-170
StableDreamer.training_step, threestudio/systems/utils.py:19-53 optimizer parsing).

One process per GPU; when torch.distributed is initialised the field-parameter gradients are mean
all-reduced once per optimizer step over RCCL (what Lightning's DDP does in the reference, launch.py:233-240)
— see scaledreamer_amd/dist.py.
"""
from __future__ import annotations

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
    if config["name"] in ("Adam", "AdamW") and "fused" not in args and "foreach" not in args and next(model.parameters()).is_cuda:
        # same update rule as the reference's torch.optim call, executed as one multi-tensor kernel per parameter group
        # instead of ~20 foreach kernels per step (each costs a launch; the hash table alone is 12.6 M entries)
        args["fused"] = True
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
        return {**self.renderer(**batch)}

    def on_train_batch_start(self, batch_idx: int = 0):
        """systems/base.py:180-184: per-step update hooks (occupancy grid, timestep annealing, resolution)."""
        self.do_update_step(self.current_epoch, self.true_global_step)
        if self.guidance is not None:
            self.guidance.do_update_step(self.current_epoch, self.true_global_step)

    def training_step(self, batch, batch_idx: int = 0):
        out = self(batch)
        guidance_out = self.guidance(out["comp_rgb"], self.prompt_utils, **batch, rgb_as_latents=False)
        loss = 0.0
        for name, value in guidance_out.items():
            self.log(f"train/{name}", value)
            if name.startswith("loss_"):
                loss = loss + value * self.C(self.cfg.loss[name.replace("loss_", "lambda_")])
        if "coarse" not in self.cfg.stage:
            raise ValueError(f"stage {self.cfg.stage!r}: only the NeRF ('coarse') stage is on the ASD hot path")
        L = self.cfg.loss
        if self.C(L.get("lambda_orient", 0.0)) > 0:
            if "normal" not in out:
                raise ValueError("Normal is required for orientation loss, no normal is found in the output.")
            loss_orient = (out["weights"].detach() * dot(out["normal"], out["t_dirs"]).clamp_min(0.0) ** 2).sum() / (out["opacity"] > 0).sum()
            self.log("train/loss_orient", loss_orient)
            loss = loss + loss_orient * self.C(L["lambda_orient"])
        if self.C(L.get("lambda_sparsity", 0.0)) > 0:
            loss_sparsity = (out["opacity"] ** 2 + 0.01).sqrt().mean()
            self.log("train/loss_sparsity", loss_sparsity)
            loss = loss + loss_sparsity * self.C(L["lambda_sparsity"])
        if self.C(L.get("lambda_opaque", 0.0)) > 0:
            oc = out["opacity"].clamp(1.0e-3, 1.0 - 1.0e-3)
            loss_opaque = binary_cross_entropy(oc, oc)
            self.log("train/loss_opaque", loss_opaque)
            loss = loss + loss_opaque * self.C(L["lambda_opaque"])
        if self.C(L.get("lambda_z_variance", 0.0)) > 0:
            if "z_variance" not in out:
                raise ValueError("z_variance is required for z_variance loss, no z_variance is found in the output.")
            loss_z = out["z_variance"][out["opacity"] > 0.5].mean()
            self.log("train/loss_z_variance", loss_z)
            loss = loss + loss_z * self.C(L["lambda_z_variance"])
        return {"loss": loss}

    def train_one_step(self, batch) -> torch.Tensor:
        """on_train_batch_start -> training_step -> backward -> (all-reduce) -> optimizer.step."""
        self.on_train_batch_start()
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.training_step(batch)["loss"]
        loss.backward()
        asd_dist.allreduce_mean_grads(self.optimizer)
        self.optimizer.step()
        self.true_global_step += 1
        return loss.detach()
