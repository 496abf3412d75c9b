"""TEST INFRASTRUCTURE — torch restatement of the Adan update of threestudio/systems/optimizers.py:200-315 (one parameter group,
global-norm clipping), pinned by tests/golden/adan_steps.npz (produced by the reference's own optimizer class).  The product's Adan
(scaledreamer_amd/optimizers.py) runs this update as a fused HIP kernel and is compared with this file / the golden on the GPU."""
import math

import torch


@torch.no_grad()
def adan_step(groups, t: int, betas=(0.98, 0.92, 0.99), eps=1e-8, weight_decay=0.0, max_grad_norm=0.0, no_prox=False):
    """groups: list of dict(params=[...], grads=[...], state=[{}...], lr=float); in-place update.  The clipping factor is global over
    all groups and uses the LAST group's eps (optimizers.py:243-262)."""
    b1, b2, b3 = betas
    clip = 1.0
    if max_grad_norm > 0:
        norm = torch.sqrt(sum(g.pow(2).sum() for grp in groups for g in grp["grads"]))
        clip = float(torch.clamp(max_grad_norm / (norm + eps), max=1.0))
    for grp in groups:
        lr = grp["lr"]
        for p, g, st in zip(grp["params"], grp["grads"], grp["state"]):
            if not st:
                st.update(exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p), exp_avg_diff=torch.zeros_like(p))
            if "neg_pre_grad" not in st or t == 1:
                st["neg_pre_grad"] = g.clone().mul_(-clip)
            g = g * clip
            d = st["neg_pre_grad"] + g
            st["exp_avg"].mul_(b1).add_(g, alpha=1 - b1)
            st["exp_avg_diff"].mul_(b2).add_(d, alpha=1 - b2)
            u = d * b2 + g
            st["exp_avg_sq"].mul_(b3).addcmul_(u, u, value=1 - b3)
            den = st["exp_avg_sq"].sqrt() / math.sqrt(1.0 - b3 ** t) + eps
            if no_prox:
                p.mul_(1 - lr * weight_decay)
            p.addcdiv_(st["exp_avg"], den, value=-lr / (1.0 - b1 ** t))
            p.addcdiv_(st["exp_avg_diff"], den, value=-lr * b2 / (1.0 - b2 ** t))
            if not no_prox:
                p.div_(1 + lr * weight_decay)
            st["neg_pre_grad"] = -g
