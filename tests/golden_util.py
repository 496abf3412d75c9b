"""Loading of the committed golden fixtures (tests/golden/*.npz) + the parameter generation rules that
tests/golden/make_goldens.py used (restated here so that the 12.6 M-entry tables never need committing)."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RENDERER_GOLDENS = ["renderer_c1_32x32x16", "renderer_c2mini_12x12x512"]


def grid_params(seed: int, n: int, amp: float) -> np.ndarray:
    return np.random.default_rng(seed).uniform(-amp, amp, n).astype(np.float32)


def load_renderer_golden(name: str) -> dict:
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    seed, amp = int(g["seed"]), float(g["grid_amp"])
    g["grid"] = grid_params(seed, 12_599_920, amp)
    g["bgrid"] = grid_params(seed + 1, 1_581_184, 0.5)
    g["spp"] = int(g["spp"])
    return g


def reference_loss_torch(out: dict, g: dict):
    """The scalar the golden script back-propagated (probes + scaledreamer.py:69-91 regularisers), written
    against torch tensors shaped like the reference's output dictionary."""
    import torch

    h, w = int(g["h"]), int(g["w"])
    g_rgb = torch.as_tensor(g["g_rgb"]).to(out["comp_rgb"])
    g_depth = torch.as_tensor(g["g_depth"]).to(out["comp_rgb"])
    dot = lambda a, b: (a * b).sum(-1, keepdim=True)
    loss_probe = (out["comp_rgb"].view(1, h, w, 3) * g_rgb).sum() + 0.1 * (out["depth"].view(1, h, w, 1) * g_depth).sum()
    loss_orient = (out["weights"].detach() * dot(out["normal"], out["t_dirs"]).clamp_min(0.0) ** 2).sum() / (out["opacity"] > 0).sum()
    loss_sparsity = (out["opacity"] ** 2 + 0.01).sqrt().mean()
    oc = out["opacity"].clamp(1.0e-3, 1.0 - 1.0e-3)
    loss_opaque = (-(oc * torch.log(oc) + (1 - oc) * torch.log(1 - oc))).mean()  # utils/ops.py:365-369
    m = out["opacity"] > 0.5
    loss_zvar = out["z_variance"][m].mean() if m.any() else out["z_variance"].sum() * 0
    return loss_probe + 10.0 * loss_orient + 30.0 * loss_sparsity + 5.0 * loss_opaque + 3.0 * loss_zvar
