"""`no-material` (threestudio/models/materials/no_material.py:16-65)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from .base import BaseModule
from .networks import get_activation, get_mlp
from .registry import register


class BaseMaterial(BaseModule):
    @dataclass
    class Config(BaseModule.Config):
        pass

    cfg: Config
    requires_normal: bool = False
    requires_tangent: bool = False
    reads_normal: bool = True      # does forward() use `normal` / `shading_normal`?  (False lets the renderer defer them)

    def configure(self):
        pass

    def forward(self, *args, **kwargs):
        raise NotImplementedError


@register("no-material")
class NoMaterial(BaseMaterial):
    @dataclass
    class Config(BaseMaterial.Config):
        n_output_dims: int = 3
        color_activation: str = "sigmoid"
        input_feature_dims: Optional[int] = None
        mlp_network_config: Optional[dict] = None
        requires_normal: bool = False

    cfg: Config
    reads_normal = False           # colour = activation(features): the normals are never looked at (no_material.py:41-54)

    @property
    def elementwise(self) -> bool:
        """colour rows are independent functions of feature rows with no parameters behind them: the renderer may hand over
        capacity-sized buffers whose tail rows are garbage (a colour MLP would sum its weight gradient over those rows too)"""
        return not self.use_network


    def configure(self) -> None:
        self.use_network = False
        if self.cfg.input_feature_dims is not None and self.cfg.mlp_network_config is not None:
            self.network = get_mlp(self.cfg.input_feature_dims, self.cfg.n_output_dims, self.cfg.mlp_network_config)
            self.use_network = True
        self.requires_normal = self.cfg.requires_normal

    def forward(self, features: torch.Tensor, **kwargs) -> torch.Tensor:
        if not self.use_network:
            assert features.shape[-1] == self.cfg.n_output_dims, (
                f"Expected {self.cfg.n_output_dims} output dims, only got {features.shape[-1]} dims input."
            )
            return get_activation(self.cfg.color_activation)(features)
        color = self.network(features.view(-1, features.shape[-1])).view(*features.shape[:-1], self.cfg.n_output_dims)
        return get_activation(self.cfg.color_activation)(color)
