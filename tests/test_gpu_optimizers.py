"""The optimizer step on the HIP path (csrc/optim.hip): the fused multi-tensor AdamW / Adam against torch.optim on the same device
tensors, the fused Adan against the golden of the reference's own optimizer class (tests/golden/adan_steps.npz) and the oracle."""
import numpy as np
import pytest
import torch

from test_host_logic_cpu import adan_cases

pytestmark = pytest.mark.gpu


def _same_update(got, want):
    """same arithmetic, not the same instruction stream (ATen contracts some multiply-adds, its foreach path rounds beta2 * v before
    the addcmul): elementwise agreement to a few ulps, with the tolerance scaled to the tensor (entries near zero after cancellation)"""
    scale = float(want.abs().max())
    torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-6 * max(scale, 1e-3))


@pytest.mark.parametrize("name,kw", [("AdamW", dict(betas=(0.0, 0.99), eps=1e-15)), ("AdamW", dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)),
                                     ("Adam", dict(betas=(0.9, 0.99), eps=1e-15)), ("Adam", dict(betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01))])
def test_fused_adamw_matches_torch(name, kw):
    from scaledreamer_amd.optimizers import AdamW

    torch.manual_seed(0)
    shapes = [(1_000_003,), (64, 32), (1, 64), (3, 64), (7,), (4099,)] + [(17, 5)] * 30          # > 24 tensors: two launches
    ref_p = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    our_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p]
    groups = lambda ps: [{"params": ps[:1], "lr": 1e-2}, {"params": ps[1:4], "lr": 1e-3}, {"params": ps[4:], "lr": 3e-3}]
    ref = getattr(torch.optim, name)(groups(ref_p), **kw)
    ours = AdamW(groups(our_p), adam_l2=name == "Adam", **({"weight_decay": 0.0} if name == "Adam" and "weight_decay" not in kw else {}), **kw)
    for step in range(5):
        for a, b in zip(ref_p, our_p):
            g = torch.randn_like(a) * (10.0 ** (step - 2))
            g.view(-1)[::3] = 0.0                                   # untouched entries: still decayed / their v still decays
            a.grad, b.grad = g.clone(), g.clone()
        ref.step()
        ours.step()
        for a, b in zip(ref_p, our_p):
            _same_update(b.detach(), a.detach())
    sd = ours.state_dict()                                          # same state layout as torch's: loads into torch.optim and back
    getattr(torch.optim, name)(groups([torch.nn.Parameter(p.detach().clone()) for p in our_p]), **kw).load_state_dict(sd)
    st = ours.state[our_p[0]]
    _same_update(st["exp_avg"], ref.state[ref_p[0]]["exp_avg"])
    _same_update(st["exp_avg_sq"], ref.state[ref_p[0]]["exp_avg_sq"])
    assert int(st["step"]) == 5


@pytest.mark.parametrize("name", ["AdamW", "Adam"])
def test_torch_optimizer_state_loads_into_the_fused_one_and_steps(name):
    """the direction a reference checkpoint takes: a state dict written by torch.optim (no `adam_l2` key in its param_groups, `step`
    moved to the device by load_state_dict) is loaded into the fused optimizer, which then continues exactly like torch does"""
    from scaledreamer_amd.optimizers import AdamW

    torch.manual_seed(1)
    ref_p = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in [(4099,), (64, 32)]]
    our_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p]
    kw = dict(lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01)
    ref = getattr(torch.optim, name)(ref_p, **kw)
    grads = [[torch.randn_like(p) for p in ref_p] for _ in range(5)]
    for step in range(3):
        for p, g in zip(ref_p, grads[step]):
            p.grad = g.clone()
        ref.step()
    ours = AdamW(our_p, adam_l2=name == "Adam", **kw)
    import copy

    ours.load_state_dict(copy.deepcopy(ref.state_dict()))     # what a checkpoint file hands over (load_state_dict aliases same-device tensors)
    assert all(g["adam_l2"] == (name == "Adam") for g in ours.param_groups)
    for a, b in zip(ref_p, our_p):
        b.data.copy_(a.data)
    for step in range(3, 5):
        for a, b, g in zip(ref_p, our_p, grads[step]):
            a.grad, b.grad = g.clone(), g.clone()
        ref.step()
        ours.step()
    for a, b in zip(ref_p, our_p):
        _same_update(b.detach(), a.detach())
    assert int(ours.state[our_p[0]]["step"]) == 5 and not ours.state[our_p[0]]["step"].is_cuda


def test_fused_adan_matches_reference_golden_and_oracle():
    from oracle.adan_ref import adan_step
    from scaledreamer_amd.optimizers import Adan

    g, seeded, cases = adan_cases()
    for tag, kw in cases:
        p1, p2 = torch.nn.Parameter(seeded("adan.p1", (7, 5)).cuda()), torch.nn.Parameter(seeded("adan.p2", (11,)).cuda())
        opt = Adan([{"params": [p1], "lr": 0.01}, {"params": [p2], "lr": 0.003}], betas=(0.98, 0.92, 0.99), eps=1e-15, **kw)
        for step in range(4):
            p1.grad, p2.grad = seeded(f"adan.g1.{step}", (7, 5)).cuda(), seeded(f"adan.g2.{step}", (11,)).cuda()
            opt.step()
        np.testing.assert_allclose(p1.detach().cpu().numpy(), g[f"{tag}.p1"], rtol=2e-6, atol=2e-7)
        np.testing.assert_allclose(p2.detach().cpu().numpy(), g[f"{tag}.p2"], rtol=2e-6, atol=2e-7)
    # a larger problem against the oracle (several chunks per tensor, tail elements)
    torch.manual_seed(1)
    ps = [torch.randn(n) for n in (10_001, 4096, 33)]
    ours = [torch.nn.Parameter(p.clone().cuda()) for p in ps]
    opt = Adan(ours, lr=2e-3, betas=(0.98, 0.92, 0.99), eps=1e-8, weight_decay=0.01, max_grad_norm=1.0)
    grp = [dict(params=ps, state=[{} for _ in ps], lr=2e-3)]
    for step in range(3):
        gs = [torch.randn_like(p) for p in ps]
        grp[0]["grads"] = [x.clone() for x in gs]
        adan_step(grp, step + 1, betas=(0.98, 0.92, 0.99), eps=1e-8, weight_decay=0.01, max_grad_norm=1.0)
        for p, x in zip(ours, gs):
            p.grad = x.cuda()
        opt.step()
    for p, q in zip(ours, ps):
        torch.testing.assert_close(p.detach().cpu(), q, rtol=1e-5, atol=1e-7)
    with pytest.raises(TypeError):
        cpu = torch.nn.Parameter(torch.zeros(3))
        cpu.grad = torch.ones(3)
        Adan([cpu]).step()
