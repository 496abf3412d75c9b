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


def test_hooked_gradient_exchange_runs_inside_a_real_training_step_on_rccl(monkeypatch):
    """the product wiring of the N > 1 path — GradientExchange.prepare -> backward whose post-accumulate hooks launch async RCCL
    all-reduces (ReduceOp.AVG) from the autograd thread -> finish -> fused AdamW on gradients that are views of the flat bucket —
    inside a real tiny ASD step (smoke configuration), against the same step without the exchange: identical parameters."""
    import torch.distributed as dist

    from scaledreamer_amd import dist as asd_dist
    from scaledreamer_amd.smoke import build_smoke_system

    torch.cuda.set_device(0)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    def run(n_steps, exchange):
        import random

        system, batches = build_smoke_system(seed=7, n_batches=n_steps)
        losses = []
        for b in batches:
            torch.manual_seed(100 + len(losses))             # same noise / t / jitter / background draws in both runs
            random.seed(100 + len(losses))
            losses.append(float(system.train_one_step(b)))
        ex = system.gradient_exchange()
        assert (ex is not None) == exchange
        if exchange:
            assert ex.prepare_called == n_steps and ex._order_learned and sorted(ex.order) == list(range(len(ex.units)))
            assert ex.exposed_ms() >= 0.0
            assert any(u["flat"] is None for u in ex.units) and any(u["flat"] is not None for u in ex.units)
        return losses, [p.grad.detach().clone() for p in system.parameters() if p.grad is not None]

    run(1, exchange=False)                                         # first contact: tunes the GEMM plans of this configuration's shapes
    ref_losses, ref_grads = run(1, exchange=False)
    ref_losses2, ref_grads2 = run(1, exchange=False)                         # run-to-run spread of the step itself (fp32 atomics order upstream of a
                                                                    # random-weight UNet: ~1 % of the largest hash-table gradient entry)
    ref_losses3, ref_grads3 = run(1, exchange=False)                        # three runs: the largest pairwise distance estimates the spread (with two,
                                                                    # |a - b| <= 3 |b - b2| fails a few percent of the time for identical distributions)
    with tempfile.TemporaryDirectory() as d:
        dist.init_process_group(backend="nccl", init_method=f"file://{d}/rdv", rank=0, world_size=1)
        try:
            monkeypatch.setattr(asd_dist, "is_distributed", lambda: True)
            losses, grads = run(1, exchange=True)
            run(3, exchange=True)                                 # and it keeps running (order learned in step 1, reused afterwards)
            torch.cuda.synchronize()
        finally:
            dist.destroy_process_group()
    # the loss of a step is not bit-reproducible either (order of the fp32 LDS atomics of the GroupNorm statistics records, amplified by
    # a random-weight UNet: a few 1e-4 relative between identical runs), so it is held to the measured spread as well
    loss_spread = max(abs(ref_losses[0] - ref_losses2[0]), abs(ref_losses[0] - ref_losses3[0]), abs(ref_losses2[0] - ref_losses3[0]))
    assert abs(losses[0] - ref_losses[0]) <= 4.0 * loss_spread + 2e-3 * abs(ref_losses[0])
    assert len(grads) == len(ref_grads)
    for a, b, b2, b3 in zip(grads, ref_grads, ref_grads2, ref_grads3):
        # mean over one rank == the local gradient of the first step, up to the step's own run-to-run spread.  Later steps are not
        # compared: AdamW with betas (0, 0.99) turns the sign of a noise-level gradient entry into a +-lr step
        spread = max(float((b - b2).abs().max()), float((b - b3).abs().max()), float((b2 - b3).abs().max()))
        # (a wrong reduction — a sum taken for a mean, a bucket exchanged twice, a stale view — moves entries by ~max|b|, far outside this)
        assert float((a - b).abs().max()) <= 4.0 * spread + 2e-2 * float(b.abs().max()) + 1e-12


def test_library_communicator_mean_allreduce_single_rank_through_the_c_abi():
    """include/asd_hip.h asd_comm_* / asd_allreduce_mean_f32 with bare ctypes: a one-rank communicator on this device, the in-place
    mean of a large buffer on a side stream (identity for one rank, bit for bit), error codes for bad arguments."""
    import ctypes as C

    from scaledreamer_amd import _lib

    torch.cuda.set_device(0)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    lib = _lib.lib()
    uid = (C.c_char * 128)()
    assert lib.asd_comm_unique_id(uid) == 0, lib.asd_last_error()
    comm = C.c_void_p()
    assert lib.asd_comm_create(uid, C.c_int32(0), C.c_int32(1), C.byref(comm)) == 0, lib.asd_last_error()
    try:
        x = torch.randn(12_599_920, device="cuda")                   # the hash table gradient of the headline configuration: 50 MB
        want = x.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        assert lib.asd_allreduce_mean_f32(comm, C.c_void_p(x.data_ptr()), C.c_int64(x.numel()), C.c_void_p(side.cuda_stream)) == 0
        assert lib.asd_allreduce_mean_f32(comm, C.c_void_p(x.data_ptr()), C.c_int64(0), C.c_void_p(side.cuda_stream)) == 0
        side.synchronize()
        assert torch.equal(x, want)
        assert lib.asd_allreduce_mean_f32(None, C.c_void_p(x.data_ptr()), C.c_int64(4), None) != 0
        assert lib.asd_comm_create(uid, C.c_int32(1), C.c_int32(1), C.byref(C.c_void_p())) != 0       # rank outside the world
    finally:
        assert lib.asd_comm_destroy(comm) == 0


def test_gradient_exchange_on_the_library_communicator(monkeypatch):
    """ASD_OWN_ALLREDUCE=1: GradientExchange launches its units through asd_allreduce_mean_f32 (side stream, stream hand-offs) instead
    of torch.distributed; one rank, so the gradients come back unchanged and the optimizer sees them."""
    import torch.distributed as dist

    from scaledreamer_amd import dist as asd_dist

    torch.cuda.set_device(0)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    monkeypatch.setenv("ASD_OWN_ALLREDUCE", "1")
    with tempfile.TemporaryDirectory() as d:
        dist.init_process_group(backend="nccl", init_method=f"file://{d}/rdv", rank=0, world_size=1)
        try:
            monkeypatch.setattr(asd_dist, "is_distributed", lambda: True)
            big = torch.nn.Parameter(torch.randn(2 << 20, device="cuda"))
            small = [torch.nn.Parameter(torch.randn(64, 32, device="cuda")), torch.nn.Parameter(torch.randn(7, device="cuda"))]
            opt = torch.optim.SGD([{"params": [big]}, {"params": small}], lr=1.0)
            ex = asd_dist.GradientExchange([big] + small)
            assert ex._own is not None
            for step in range(3):
                ex.prepare()
                loss = (big * big).sum() * 0.5 + sum((p * p).sum() * 0.5 for p in small)        # d/dp = p
                want = [p.detach().clone() for p in [big] + small]
                loss.backward()
                ex.finish()
                torch.cuda.synchronize()
                for p, w in zip([big] + small, want):
                    assert torch.equal(p.grad.view_as(w), w)
            ex.close()
        finally:
            dist.destroy_process_group()
