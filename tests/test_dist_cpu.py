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
