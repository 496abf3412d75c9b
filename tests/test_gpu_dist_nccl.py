"""The data-parallel exchange step on the real backend (nccl = RCCL on ROCm).  The GPU box has ONE device, so the process group has
a single rank: this checks that the calls the N > 1 path makes — in-place async all-reduce with ReduceOp.AVG on a large gradient,
the flat bucket of the small ones, parameter broadcast — are accepted by RCCL and leave the values intact; the arithmetic across
ranks is covered on CPU with gloo, world_size 2 (tests/test_dist_cpu.py)."""
import os
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_allreduce_mean_grads_runs_on_rccl_single_rank(monkeypatch):
    import torch.distributed as dist

    from scaledreamer_amd import dist as asd_dist

    torch.cuda.set_device(0)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with tempfile.TemporaryDirectory() as d:
        dist.init_process_group(backend="nccl", init_method=f"file://{d}/rdv", rank=0, world_size=1)
        try:
            monkeypatch.setattr(asd_dist, "is_distributed", lambda: True)
            big = torch.nn.Parameter(torch.zeros(2 << 20, device="cuda"))        # 8 MB: reduced in place
            small = [torch.nn.Parameter(torch.zeros(64, 32, device="cuda")), torch.nn.Parameter(torch.zeros(7, device="cuda"))]
            gen = torch.Generator(device="cuda").manual_seed(3)
            for p in [big] + small:
                p.grad = torch.randn(p.shape, device="cuda", generator=gen)
            want = [p.grad.clone() for p in [big] + small]
            opt = torch.optim.AdamW([{"params": [big]}, {"params": small}], lr=1e-3)
            asd_dist.allreduce_mean_grads(opt)
            torch.cuda.synchronize()
            for p, w in zip([big] + small, want):
                assert torch.equal(p.grad, w)                                     # mean over one rank

            lin = torch.nn.Linear(8, 8).cuda()
            before = [p.detach().clone() for p in lin.parameters()]
            asd_dist.broadcast_parameters(lin)
            for p, w in zip(lin.parameters(), before):
                assert torch.equal(p, w)
            dist.barrier()
        finally:
            dist.destroy_process_group()
