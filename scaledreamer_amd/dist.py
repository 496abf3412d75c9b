"""Data-parallel gradient exchange: one process per GPU, mean all-reduce of every trainable-parameter
gradient once per optimizer step (what Lightning DDP does for the reference, launch.py:233-240).

Backend "nccl" is RCCL on ROCm (xGMI inside a node); "gloo" is used by the CPU tests.  The gradients of one
step are flattened into few large buckets (the hash table is 50 MB on its own) so that each all-reduce is
one large RCCL call — per-link-bound ring traffic favours few big collectives over many small ones.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import sys
from typing import List

import torch
import torch.distributed as dist

BUCKET_BYTES = 256 << 20
IN_PLACE_BYTES = 4 << 20   # gradients at least this large are all-reduced in place


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


@contextlib.contextmanager
def stdout_to_stderr():
    """RCCL prints a version banner with printf to the C stdout of every rank (it surfaces when the buffer is flushed, i.e. at
    exit, AFTER anything Python printed).  bench.py's contract is one JSON line on stdout, so communicator creation and teardown
    run with file descriptor 1 pointing at stderr and the C buffers are flushed before it is restored."""
    sys.stdout.flush()
    libc = ctypes.CDLL(None)
    libc.fflush(None)
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


def init_from_env(backend: str = None) -> int:
    """Initialise the process group from torchrun's environment (RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or dist.is_initialized():
        return world
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    with stdout_to_stderr():
        dist.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=world)
        if backend == "nccl":   # communicators are created lazily: force it (and the banner) now
            t = torch.zeros(1, device="cuda")
            dist.all_reduce(t)
            torch.cuda.synchronize()
    return world


def shutdown() -> None:
    if dist.is_available() and dist.is_initialized():
        with stdout_to_stderr():
            dist.destroy_process_group()


def _buckets(grads: List[torch.Tensor]):
    cur, size = [], 0
    for g in grads:
        nb = g.numel() * g.element_size()
        if cur and size + nb > BUCKET_BYTES:
            yield cur
            cur, size = [], 0
        cur.append(g)
        size += nb
    if cur:
        yield cur


def allreduce_mean_grads(optimizer: torch.optim.Optimizer) -> None:
    """mean of every gradient over the ranks.  Large contiguous gradients (the 50 MB hash table) are reduced in place — no
    flatten / scatter copies —, the small ones travel together in flat buckets; all calls are issued asynchronously and
    waited for once, so RCCL can pipeline them over the xGMI links."""
    if not is_distributed():
        return
    world = dist.get_world_size()
    grads = [p.grad for grp in optimizer.param_groups for p in grp["params"] if p.grad is not None]
    big = [g for g in grads if g.is_contiguous() and g.numel() * g.element_size() >= IN_PLACE_BYTES]
    small = [g for g in grads if not any(g is b for b in big)]
    avg = dist.ReduceOp.AVG if dist.get_backend() == "nccl" else dist.ReduceOp.SUM
    works, flats = [], []
    for g in big:
        works.append(dist.all_reduce(g.view(-1), op=avg, async_op=True))
    for bucket in _buckets(small):
        flat = torch.cat([g.reshape(-1) for g in bucket])
        flats.append((flat, bucket))
        works.append(dist.all_reduce(flat, op=avg, async_op=True))
    for w in works:
        w.wait()
    if avg == dist.ReduceOp.SUM:
        for g in big:
            g.div_(world)
    for flat, bucket in flats:
        if avg == dist.ReduceOp.SUM:
            flat.div_(world)
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """identical initial parameters on every rank (DDP broadcasts from rank 0 at wrap time)."""
    if not is_distributed():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        if t.numel() == 0:
            continue
        if t.dtype == torch.bool:  # e.g. the occupancy grid's `binaries`
            u = t.data.to(torch.uint8)
            dist.broadcast(u, src=src)
            t.data.copy_(u.bool())
        else:
            dist.broadcast(t.data, src=src)
