"""`neural-environment-map-background`
(threestudio/models/background/neural_environment_map_background.py:15-67) on the HIP path."""
from __future__ import annotations

import random
from dataclasses import dataclass, field
from typing import Optional, Tuple

import torch

from . import ops
from .base import BaseModule
from .networks import VanillaMLP, get_activation, get_encoding, get_mlp
from .registry import register


class BaseBackground(BaseModule):
    @dataclass
    class Config(BaseModule.Config):
        pass

    cfg: Config

    def configure(self):
        pass

    def forward(self, dirs: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError


class _EnvMapFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dirs, grid, w0, w1, w2, meta):
        color = ops.envmap_fwd(meta, grid, w0, w1, w2, dirs)
        ctx.save_for_backward(dirs, grid, w0, w1, w2)
        ctx.meta = meta
        return color

    @staticmethod
    def backward(ctx, d_color):
        dirs, grid, w0, w1, w2 = ctx.saved_tensors
        dgrid, dw0, dw1, dw2 = ops.envmap_bwd(ctx.meta, grid, w0, w1, w2, dirs, d_color.contiguous())
        return None, dgrid, dw0, dw1, dw2, None


@register("neural-environment-map-background")
class NeuralEnvironmentMapBackground(BaseBackground):
    @dataclass
    class Config(BaseBackground.Config):
        n_output_dims: int = 3
        color_activation: str = "sigmoid"
        dir_encoding_config: dict = field(default_factory=lambda: {"otype": "SphericalHarmonics", "degree": 3})
        mlp_network_config: dict = field(
            default_factory=lambda: {"otype": "VanillaMLP", "activation": "ReLU", "n_neurons": 16, "n_hidden_layers": 2}
        )
        random_aug: bool = False
        random_aug_prob: float = 0.5
        eval_color: Optional[Tuple[float, float, float]] = None

    cfg: Config

    def configure(self) -> None:
        self.encoding = get_encoding(3, self.cfg.dir_encoding_config)
        self.network = get_mlp(self.encoding.n_output_dims, self.cfg.n_output_dims, self.cfg.mlp_network_config)
        m = self.cfg.mlp_network_config
        self._meta = getattr(self.encoding.encoding.encoding, "meta", None)   # None: parameter-free encodings (SphericalHarmonics)
        self._fused = (
            self._meta is not None and self._meta.n_levels == 4 and not self.encoding.include_xyz and isinstance(self.network, VanillaMLP)
            and m.get("n_neurons") == 16 and m.get("n_hidden_layers") == 2 and self.cfg.n_output_dims == 3
            and self.cfg.color_activation == "sigmoid" and m.get("output_activation", "none") in (None, "none")
        )
        # injectable RNG hooks: tests replace these to pin "identical inputs"
        self.rand_fn = random.random
        self.rand_color_fn = lambda b, c: torch.rand(b, 1, 1, c)

    def forward(self, dirs: torch.Tensor) -> torch.Tensor:
        if not self.training and self.cfg.eval_color is not None:
            return torch.ones(*dirs.shape[:-1], self.cfg.n_output_dims).to(dirs) * torch.as_tensor(self.cfg.eval_color).to(dirs)
        if self._fused and dirs.is_cuda:
            flat = dirs.reshape(-1, 3).contiguous().float()
            grid = self.encoding.encoding.encoding.params
            w0, w1, w2 = (self.network.layers[i].weight for i in (0, 2, 4))
            if torch.is_grad_enabled() and grid.requires_grad:
                color = _EnvMapFn.apply(flat, grid, w0, w1, w2, self._meta)
            else:
                color = ops.envmap_fwd(self._meta, grid.detach(), w0.detach(), w1.detach(), w2.detach(), flat)
            color = color.view(*dirs.shape[:-1], 3)
        else:
            d01 = (dirs + 1.0) / 2.0
            color = self.network(self.encoding(d01.view(-1, 3))).view(*dirs.shape[:-1], self.cfg.n_output_dims)
            color = get_activation(self.cfg.color_activation)(color)
        if self.training and self.cfg.random_aug and self.rand_fn() < self.cfg.random_aug_prob:
            # random solid colour; `color * 0 +` keeps every parameter in the autograd graph (DDP)
            rc = self.rand_color_fn(dirs.shape[0], self.cfg.n_output_dims)      # host RNG, as in the reference (its draw order is the contract)
            if dirs.is_cuda and not rc.is_cuda:
                # through pinned memory, asynchronously: a pageable .to(device) blocks the host until the stream reaches the copy —
                # every second step (random_aug_prob = 0.5) the host lost the lead it has over the GPU
                rc = rc.pin_memory().to(dirs.device, non_blocking=True)
            color = color * 0 + rc.to(dirs).expand(*dirs.shape[:-1], -1)
        return color
