"""The N > 1 path on CPU: two gloo processes, per-rank seeds, mean all-reduce of the gradients
(the DDP contract of the reference: launch.py:171,233-240)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from scaledreamer_amd import dist as asd_dist

    assert asd_dist.init_from_env("gloo") == world
    torch.manual_seed(10 + rank)                     # per-rank seed = cfg.seed + rank
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    big = torch.nn.Parameter(torch.randn(300_000))   # stands for the 12.6 M-entry hash table (own bucket)
    model.register_parameter("table", big)
    asd_dist.broadcast_parameters(model)             # identical initial parameters on every rank
    ref = [p.detach().clone() for p in model.parameters()]
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2)
    x = torch.randn(5, 8)                             # different data per rank
    loss = model(x).pow(2).mean() + (big[:1000] * (rank + 1)).sum()
    loss.backward()
    local = [p.grad.clone() for p in model.parameters()]
    old_bucket = asd_dist.BUCKET_BYTES
    asd_dist.BUCKET_BYTES = 4096                     # force several buckets incl. multi-tensor ones
    asd_dist.IN_PLACE_BYTES = 1 << 20                # the 1.2 MB "table" takes the in-place path of the hash-grid gradient
    asd_dist.allreduce_mean_grads(opt)
    asd_dist.BUCKET_BYTES = old_bucket
    torch.save({"init": ref, "local": local, "avg": [p.grad.clone() for p in model.parameters()]}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_mean_allreduce(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    for a, b in zip(r0["init"], r1["init"]):
        assert torch.equal(a, b), "parameters were not broadcast from rank 0"
    for g0, g1, a0, a1 in zip(r0["local"], r1["local"], r0["avg"], r1["avg"]):
        assert not torch.equal(g0, g1) or g0.abs().sum() == 0
        torch.testing.assert_close(a0, (g0 + g1) / 2)
        torch.testing.assert_close(a1, a0)


def _own_worker(rank, world, port, out_dir):
    """GradientExchange on a collective object of its own (the ASD_OWN_ALLREDUCE=1 path: OwnCollective over asd_allreduce_mean_f32 on GPU ranks) —
    here a stand-in with the same interface over gloo that logs what the exchange asks of it: every unit must go through launch() in the
    fixed order both ranks share (learned in step 1, reused in step 2), wait() must come after the last launch of a step and before
    finish() returns, close() with the exchange."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from scaledreamer_amd import dist as asd_dist

    assert asd_dist.init_from_env("gloo") == world

    class FakeOwn:
        def __init__(self):
            self.log, self.works = [], []

        def launch(self, flat):
            assert flat.dtype == torch.float32 and flat.is_contiguous() and flat.dim() == 1
            self.log.append(("launch", flat.numel()))
            self.works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))     # (the exchange divides by the world size on gloo)

        def wait(self):
            self.log.append(("wait", len(self.works)))
            for w in self.works:
                w.wait()
            self.works = []

        def close(self):
            self.log.append(("close", 0))

    torch.manual_seed(3 + rank)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    model.register_parameter("table", torch.nn.Parameter(torch.randn(300_000)))
    asd_dist.broadcast_parameters(model)
    own = FakeOwn()
    ex = asd_dist.GradientExchange(list(model.parameters()), bucket_bytes=256, in_place_bytes=1 << 20, own=own)
    steps = []
    for step in range(2):
        x = torch.randn(5, 8)
        loss = model(x).pow(2).mean() + (model.table[:1000] * (rank + 1 + step)).sum()
        local = torch.autograd.grad(loss, list(model.parameters()), retain_graph=True)
        ex.prepare()
        n0 = len(own.log)
        loss.backward()
        launched_in_backward = sum(1 for op, _ in own.log[n0:] if op == "launch")
        ex.finish()
        steps.append({"local": [g.clone() for g in local], "avg": [p.grad.clone() for p in model.parameters()], "log": list(own.log[n0:]),
                      "launched_in_backward": launched_in_backward})
    n_units = len(ex.units)
    ex.close()
    torch.save({"steps": steps, "n_units": n_units, "closed": own.log[-1][0] == "close", "order": ex.order}, os.path.join(out_dir, f"o{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchange_on_an_own_collective(tmp_path):
    port = 29500 + ((os.getpid() + 17) % 2000)
    mp.spawn(_own_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "o0.pt"), torch.load(tmp_path / "o1.pt")
    assert r0["n_units"] == r1["n_units"] >= 3 and r0["closed"] and r1["closed"] and r0["order"] == r1["order"]
    for step in range(2):
        s0, s1 = r0["steps"][step], r1["steps"][step]
        for log in (s0["log"], s1["log"]):
            ops = [op for op, _ in log]
            assert ops.count("launch") == r0["n_units"] and ops.count("wait") == 1 and ops[-1] == "wait", ops     # every unit, then ONE wait, nothing after it
            assert log[-1][1] == r0["n_units"]                                                                    # ... which covered all of them
        assert [n for op, n in s0["log"] if op == "launch"] == [n for op, n in s1["log"] if op == "launch"]       # same collectives in the same order
        for g0, g1, a0, a1 in zip(s0["local"], s1["local"], s0["avg"], s1["avg"]):
            torch.testing.assert_close(a0, (g0 + g1) / 2)
            torch.testing.assert_close(a1, a0)
    # step 2 uses the learned order: units are launched from the backward hooks, not held back for finish()
    assert r0["steps"][1]["launched_in_backward"] >= 1 and r1["steps"][1]["launched_in_backward"] >= 1


def _system_worker(rank, world, port, out_dir):
    """the REAL StableDreamer.train_one_step wiring (update hooks -> GradientExchange.prepare -> training_step -> backward with
    the exchange's hooks -> finish -> optimizer.step -> end hooks) with a stub renderer / guidance: per-rank seed, broadcast
    initial parameters, a parameter that gets no gradient on rank 1, launch order learned in step 1 and reused in step 2."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from scaledreamer_amd import dist as asd_dist
    from scaledreamer_amd.base import Updateable
    from scaledreamer_amd.config import ConfigDict
    from scaledreamer_amd.system import StableDreamer

    assert asd_dist.init_from_env("gloo") == world
    asd_dist.IN_PLACE_BYTES = 1 << 20
    seed = 10 + rank                                   # launch.py:171: cfg.seed + rank
    torch.manual_seed(seed)

    class Renderer(torch.nn.Module, Updateable):
        def __init__(self):
            super().__init__()
            self.table = torch.nn.Parameter(torch.randn(300_000))       # > IN_PLACE_BYTES: its own in-place unit
            self.mlp = torch.nn.Sequential(torch.nn.Linear(3, 8), torch.nn.ReLU(), torch.nn.Linear(8, 3))
            self.sometimes_unused = torch.nn.Parameter(torch.randn(7))
            self.never_used = torch.nn.Parameter(torch.randn(5))         # no rank ever produces a gradient for it
            self.register_buffer("grid_bits", torch.rand(64) > 0.5)       # bool buffer (the occupancy grid's `binaries`)
            self.updates = []

        def update_step(self, epoch, global_step, on_load_weights=False):
            self.updates.append(global_step)

        def forward(self, rays_d, **kw):
            rgb = torch.sigmoid(self.mlp(rays_d) + self.table[:3])
            if rank == 0:                                                # rank 1 produces no gradient for this parameter
                rgb = rgb + 0.1 * self.sometimes_unused[:3]
            return {"comp_rgb": rgb, "opacity": rgb.mean(-1, keepdim=True).clamp(0, 1)}

    class Guidance(Updateable):
        calls = 0

        def update_step(self, epoch, global_step, on_load_weights=False):
            Guidance.calls += 1

        def __call__(self, rgb, prompt_utils, rgb_as_latents=False, **batch):
            return {"loss_asd": (rgb ** 2).sum(), "grad_norm": rgb.detach().norm()}

    s = object.__new__(StableDreamer)
    torch.nn.Module.__init__(s)
    s.cfg = ConfigDict(stage="coarse", loss=ConfigDict(lambda_asd=1.0, lambda_orient=0.0, lambda_sparsity=2.0, lambda_opaque=0.0,
                                                        lambda_z_variance=0.0))
    s.current_epoch, s.true_global_step, s.logged = 0, 0, {}
    s.renderer, s.guidance, s.prompt_utils = Renderer(), Guidance(), None
    asd_dist.broadcast_parameters(s)
    init = {k: v.clone() for k, v in s.state_dict().items()}
    s.optimizer = torch.optim.SGD([{"params": [s.renderer.table]}, {"params": list(s.renderer.mlp.parameters()) + [s.renderer.sometimes_unused, s.renderer.never_used]}], lr=0.5)
    local, untouched_grad_is_none = [], []
    for step in range(2):
        batch = {"rays_d": torch.randn(1, 4, 4, 3)}    # different data per rank (torch.manual_seed(seed) above)
        # this rank's own gradient, computed on the side
        probe = s.training_step(batch)["loss"]
        gs = torch.autograd.grad(probe, [p for p in s.renderer.parameters()], allow_unused=True)
        local.append([torch.zeros_like(p) if g is None else g.clone() for p, g in zip(s.renderer.parameters(), gs)])
        before = [p.detach().clone() for p in s.renderer.parameters()]
        s.train_one_step(batch)
        untouched_grad_is_none.append(s.renderer.never_used.grad is None and s.renderer.sometimes_unused.grad is not None)
        applied = [(b - p.detach()) / 0.5 for b, p in zip(before, s.renderer.parameters())]   # SGD: the averaged gradient
        local[-1] = (local[-1], applied)
    ex = s.gradient_exchange()
    torch.save({"init": init, "steps": local, "order": ex.order, "n_units": len(ex.units), "updates": s.renderer.updates,
                "guidance_updates": Guidance.calls, "untouched_none": untouched_grad_is_none, "final": [p.detach().clone() for p in s.renderer.parameters()]},
               os.path.join(out_dir, f"s{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_real_system_wiring(tmp_path):
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_system_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "s0.pt"), torch.load(tmp_path / "s1.pt")
    for k in r0["init"]:
        assert torch.equal(r0["init"][k], r1["init"][k]), f"{k} was not broadcast from rank 0"
    assert r0["order"] == r1["order"] and sorted(r0["order"]) == list(range(r0["n_units"])) and r0["n_units"] == 2
    assert r0["updates"] == [0, 1] and r0["guidance_updates"] == 2          # hooks once per step, guidance not updated twice
    # a parameter NO rank touched keeps grad = None on every rank (the optimizer skips it as in a single process); one that only
    # rank 0 touched gets the mean everywhere
    assert r0["untouched_none"] == [True, True] and r1["untouched_none"] == [True, True]
    for (g0, a0), (g1, a1) in zip(r0["steps"], r1["steps"]):
        for x0, x1, y0, y1 in zip(g0, g1, a0, a1):
            torch.testing.assert_close(y0, (x0 + x1) / 2, rtol=1e-4, atol=1e-6)   # what the optimizer applied = mean over ranks
            torch.testing.assert_close(y1, y0)
        assert g1[1].abs().sum() == 0 and g0[1].abs().sum() > 0              # `sometimes_unused` (parameters(): table, sometimes_unused, never_used, mlp...): rank 1 never touched it, it still gets rank 0's half
    for a, b in zip(r0["final"], r1["final"]):
        torch.testing.assert_close(a, b)                                      # replicas stay identical


def _accumulate_worker(rank, world, port, out_dir):
    """accumulate_grad_batches = 2 through the REAL train_one_step: two micro-batches per optimizer step on each of two ranks; what the
    optimizer applies must be the mean over ranks AND micro-batches of the per-batch gradients (Lightning: loss / k per backward, DDP
    no_sync() on all but the last; configs/multi-prompt_benchmark/asd_mv_triplane_transformer_10k.yaml:129) — i.e. one step on the
    concatenated batch of a mean-reduced loss."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from scaledreamer_amd import dist as asd_dist
    from scaledreamer_amd.base import Updateable
    from scaledreamer_amd.config import ConfigDict
    from scaledreamer_amd.system import StableDreamer

    if world > 1:
        assert asd_dist.init_from_env("gloo") == world
    asd_dist.IN_PLACE_BYTES = 1 << 20
    torch.manual_seed(10 + rank)

    class Renderer(torch.nn.Module, Updateable):
        def __init__(self):
            super().__init__()
            self.table = torch.nn.Parameter(torch.randn(300_000))
            self.mlp = torch.nn.Sequential(torch.nn.Linear(3, 8), torch.nn.ReLU(), torch.nn.Linear(8, 3))
            self.second_batch_only = torch.nn.Parameter(torch.randn(3))       # no gradient in the first micro-batch of a step
            self.updates, self.calls = [], 0

        def update_step(self, epoch, global_step, on_load_weights=False):
            self.updates.append(global_step)

        def forward(self, rays_d, **kw):
            rgb = torch.sigmoid(self.mlp(rays_d) + self.table[:3])
            if self.calls % 2 == 1:
                rgb = rgb + 0.1 * self.second_batch_only
            self.calls += 1
            return {"comp_rgb": rgb, "opacity": rgb.mean(-1, keepdim=True).clamp(0, 1)}

    class Guidance(Updateable):
        def __call__(self, rgb, prompt_utils, rgb_as_latents=False, **batch):
            return {"loss_asd": (rgb ** 2).mean()}

    s = object.__new__(StableDreamer)
    torch.nn.Module.__init__(s)
    s.cfg = ConfigDict(stage="coarse", loss=ConfigDict(lambda_asd=1.0, lambda_orient=0.0, lambda_sparsity=2.0, lambda_opaque=0.0,
                                                        lambda_z_variance=0.0))
    s.current_epoch, s.true_global_step, s.logged = 0, 0, {}
    s.renderer, s.guidance, s.prompt_utils = Renderer(), Guidance(), None
    s.accumulate_grad_batches = 2
    asd_dist.broadcast_parameters(s)
    params = list(s.renderer.parameters())
    s.optimizer = torch.optim.SGD([{"params": [s.renderer.table]}, {"params": params[1:]}], lr=0.5)
    steps = []
    for step in range(2):
        before = [p.detach().clone() for p in params]
        local = [torch.zeros_like(p) for p in params]
        for micro in range(2):
            batch = {"rays_d": torch.randn(1, 4, 4, 3)}
            calls = s.renderer.calls
            gs = torch.autograd.grad(s.training_step(batch)["loss"], params, allow_unused=True)      # this batch's own gradient, on the side
            s.renderer.calls = calls
            for acc, g in zip(local, gs):
                if g is not None:
                    acc += g
            s.train_one_step(batch)
            if micro == 0:
                assert all(torch.equal(b, p.detach()) for b, p in zip(before, params)), "the optimizer stepped before the k-th batch"
                assert s.true_global_step == step
        steps.append((local, [(b - p.detach()) / 0.5 for b, p in zip(before, params)]))
    ex = s.gradient_exchange()
    torch.save({"steps": steps, "updates": s.renderer.updates, "global_step": s.true_global_step,
                "exchanges": None if ex is None else ex.prepare_called, "final": [p.detach().clone() for p in params]},
               os.path.join(out_dir, f"a{rank}.pt"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_accumulate_grad_batches(tmp_path):
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_accumulate_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "a0.pt"), torch.load(tmp_path / "a1.pt")
    assert r0["updates"] == [0, 0, 1, 1] and r0["global_step"] == 2          # update hooks per batch, global_step per optimizer step
    assert r0["exchanges"] == 4                                               # prepare() per batch; collectives only in every second one
    for (g0, a0), (g1, a1) in zip(r0["steps"], r1["steps"]):
        for x0, x1, y0, y1 in zip(g0, g1, a0, a1):
            torch.testing.assert_close(y0, (x0 + x1) / 4, rtol=1e-4, atol=1e-6)   # mean over 2 ranks x 2 micro-batches
            torch.testing.assert_close(y1, y0)
    for a, b in zip(r0["final"], r1["final"]):
        torch.testing.assert_close(a, b)


def test_single_process_accumulate_grad_batches(tmp_path):
    _accumulate_worker(0, 1, 0, str(tmp_path))
    r = torch.load(tmp_path / "a0.pt")
    assert r["updates"] == [0, 0, 1, 1] and r["global_step"] == 2 and r["exchanges"] is None
    for g, a in r["steps"]:
        for x, y in zip(g, a):
            torch.testing.assert_close(y, x / 2, rtol=1e-4, atol=1e-6)


def test_single_process_is_a_noop():
    sys.path.insert(0, ROOT)
    from scaledreamer_amd import dist as asd_dist

    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    asd_dist.allreduce_mean_grads(torch.optim.SGD([p], lr=0.1))
    assert torch.equal(p.grad, torch.full((3,), 2.0))


def test_stdout_to_stderr_keeps_c_level_prints_off_stdout():
    """RCCL printf()s a banner to the C stdout; bench.py's stdout must be exactly one JSON line (dist.stdout_to_stderr)."""
    import subprocess
    import sys

    code = (
        "import ctypes, sys\n"
        "from scaledreamer_amd import dist as D\n"
        "libc = ctypes.CDLL(None)\n"
        "with D.stdout_to_stderr():\n"
        "    libc.printf(b'BANNER from C\\n')\n"
        "    print('python print inside')\n"
        "print('{\"json\": 1}')\n"
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(__import__("pathlib").Path(__file__).resolve().parents[1]))
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == '{"json": 1}'
    assert "BANNER from C" in r.stderr and "python print inside" in r.stderr


def test_bench_two_ranks_under_torchrun_prints_one_json_line():
    """bench.py's N > 1 contract as the driver launches it (python -m torch.distributed.run ... bench.py --gpus 2), on CPU with gloo
    and the stub system (ASD_BENCH_STUB=1): rank 0 alone prints ONE JSON line on stdout, both ranks pass the barriers and exit 0, the
    gradient exchange ran in every step, the replicas stayed identical (asserted inside the script)."""
    import json
    import subprocess

    port = 33500 + (os.getpid() % 2000)
    env = dict(os.environ, ASD_BENCH_STUB="1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=240)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 2 and out["scaling"] == "weak" and out["data"] == "stub"
    assert out["exchange_units"] == 2 and out["exchange_steps"] == 6          # the table in place + one bucket; every step exchanged
    assert out["allreduce_exposed_ms"] is not None and out["value"] > 0
    # the line proves its own world: size as an all-reduce of ones saw it, one record per rank, the bytes one exchange moves
    assert out["rccl_ranks"] == 2 and [r_["rank"] for r_ in out["ranks"]] == [0, 1] and out["collective_backend"] == "gloo"
    assert out["allreduce_units"] == 2 and out["allreduce_bytes"] > 0
