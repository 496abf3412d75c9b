"""Base classes mirroring threestudio/utils/base.py:21-118 (Updateable / BaseObject / BaseModule)."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Any, Optional

import torch
import torch.nn as nn

from .config import parse_structured


def get_rank() -> int:
    for key in ("RANK", "LOCAL_RANK", "SLURM_PROCID", "JSM_NAMESPACE_RANK"):
        if os.environ.get(key) is not None:
            return int(os.environ[key])
    return 0


def get_device() -> torch.device:
    """cuda:{local rank} as the reference (utils/misc.py:29-30); the HIP path needs a GPU, but module
    construction (parameter shapes, state-dict layout) also works on a CPU-only box for the unit tests."""
    if torch.cuda.is_available():
        return torch.device(f"cuda:{int(os.environ.get('LOCAL_RANK', get_rank())) % torch.cuda.device_count()}")
    return torch.device("cpu")


class Updateable:
    def _updateable_children(self):
        """threestudio/utils/base.py:21-52 walks self.__dir__() on every call (hundreds of nn.Module attributes, each a
        getattr); the set of Updateable attributes is found the same way but cached by attribute name and re-scanned only when
        the instance's attribute names change."""
        names = self.__dict__.get("_upd_names")
        sig = (len(self.__dict__), len(getattr(self, "_modules", ())))
        if names is None or self.__dict__.get("_upd_sig") != sig:
            names = []
            for attr in self.__dir__():
                if attr.startswith("_"):
                    continue
                try:
                    module = getattr(self, attr)
                except Exception:
                    continue
                if isinstance(module, Updateable):
                    names.append(attr)
            object.__setattr__(self, "_upd_names", names)
            object.__setattr__(self, "_upd_sig", (len(self.__dict__), len(getattr(self, "_modules", ()))))
        out = []
        for attr in names:
            try:
                module = getattr(self, attr)
            except Exception:
                continue
            if isinstance(module, Updateable):
                out.append(module)
        return out

    def do_update_step(self, epoch: int, global_step: int, on_load_weights: bool = False):
        for module in self._updateable_children():
            module.do_update_step(epoch, global_step, on_load_weights=on_load_weights)
        self.update_step(epoch, global_step, on_load_weights=on_load_weights)

    def do_update_step_end(self, epoch: int, global_step: int):
        for module in self._updateable_children():
            module.do_update_step_end(epoch, global_step)
        self.update_step_end(epoch, global_step)

    def update_step(self, epoch: int, global_step: int, on_load_weights: bool = False):
        pass

    def update_step_end(self, epoch: int, global_step: int):
        pass


class BaseObject(Updateable):
    @dataclass
    class Config:
        pass

    cfg: Config

    def __init__(self, cfg: Optional[dict] = None, *args, **kwargs) -> None:
        super().__init__()
        self.cfg = parse_structured(self.Config, cfg)
        self.device = get_device()
        self.configure(*args, **kwargs)

    def configure(self, *args, **kwargs) -> None:
        pass


class BaseModule(nn.Module, Updateable):
    @dataclass
    class Config:
        weights: Optional[str] = None

    cfg: Config

    def __init__(self, cfg: Optional[dict] = None, *args, **kwargs) -> None:
        super().__init__()
        self.cfg = parse_structured(self.Config, cfg)
        self.device = get_device()
        self.configure(*args, **kwargs)
        if self.cfg.weights is not None:
            # format: path/to/weights:module_name  (utils/base.py:103-112)
            weights_path, module_name = self.cfg.weights.split(":")
            ckpt = torch.load(weights_path, map_location="cpu", weights_only=False)   # Lightning checkpoints carry non-tensor objects; the path comes from the user's own config
            prefix = module_name + "."
            sd = {k[len(prefix):]: v for k, v in ckpt["state_dict"].items() if k.startswith(prefix)}
            self.load_state_dict(sd)
            self.do_update_step(ckpt.get("epoch", 0), ckpt.get("global_step", 0), on_load_weights=True)
        self.register_buffer("_dummy", torch.zeros(0).float(), persistent=False)

    def configure(self, *args, **kwargs) -> None:
        pass
