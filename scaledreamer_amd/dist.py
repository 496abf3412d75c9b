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


class OwnCollective:
    """the exchange on the library's own communicator (include/asd_hip.h: asd_comm_* / asd_allreduce_mean_f32, RCCL resolved at run time)
    instead of torch.distributed's process group: opt-in (ASD_OWN_ALLREDUCE=1), GPU tensors only.  torch.distributed is still what carries
    the 128-byte unique id from rank 0 to the others at start-up.  All-reduces run on a side stream of their own: `launch` orders them
    behind the work already enqueued on the current stream (the kernel that produced the gradient), `wait` puts the current stream behind
    them — the same hand-offs torch's ProcessGroupNCCL makes."""

    def __init__(self, device: torch.device):
        import ctypes as C

        from . import _lib

        self._C, self._lib = C, _lib
        rank, world = dist.get_rank(), dist.get_world_size()
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_char * 128)()
            _lib.check(_lib.lib().asd_comm_unique_id(buf))
            uid = torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8).clone()
        dev_uid = uid.to(device) if dist.get_backend() == "nccl" else uid
        dist.broadcast(dev_uid, src=0)
        raw = bytes(dev_uid.cpu().tolist())
        self.comm = C.c_void_p()
        with torch.cuda.device(device), stdout_to_stderr():
            _lib.check(_lib.lib().asd_comm_create(C.c_char_p(raw), _lib.i32(rank), _lib.i32(world), C.byref(self.comm)))
        self.stream = torch.cuda.Stream(device=device)
        self._pending = False

    def launch(self, flat: torch.Tensor) -> None:
        """mean all-reduce of a contiguous fp32 tensor, in place, asynchronous"""
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
        self.stream.wait_stream(torch.cuda.current_stream(flat.device))
        flat.record_stream(self.stream)
        self._lib.check(self._lib.lib().asd_allreduce_mean_f32(self.comm, self._C.c_void_p(flat.data_ptr()), self._C.c_int64(flat.numel()),
                                                              self._C.c_void_p(self.stream.cuda_stream)))
        self._pending = True

    def wait(self) -> None:
        if self._pending:
            torch.cuda.current_stream().wait_stream(self.stream)
            self._pending = False

    def close(self) -> None:
        if self.comm:
            self._lib.lib().asd_comm_destroy(self.comm)
            self.comm = None


class GradientExchange:
    """The one exchange step of the data-parallel path (launch.py:233-240: Lightning DDP): mean of every trainable gradient.

    Design for one xGMI node (ring traffic is per-link bound, so few, large collectives; everything asynchronous):
      * the exchange is cut into UNITS in a fixed order that is identical on every rank: each gradient of at least IN_PLACE_BYTES
        (the 50 MB hash table, the feature volumes' generators) is its own unit and is all-reduced in place; the remaining
        parameters are grouped into persistent flat buckets and their `.grad`s ARE views of the bucket (autograd accumulates into
        them in place), so nothing is concatenated or copied back per step;
      * a post-accumulate hook marks a unit ready the moment its last gradient has been written by the backward pass and launches
        its all-reduce (async: RCCL orders it after the producing kernel and runs it on its own stream) while the rest of the
        backward and the optimizer preparation continue.  Units are always launched in the fixed order — a rank that gets no
        gradient for a parameter (empty-ray step, unused branch) contributes zeros from `finish()` — so the ranks can never issue
        different collectives;
      * the fixed order is the order in which rank 0 saw the units become ready in its first step (broadcast once), like DDP's
        bucket rebuild: what finishes first is sent first.
    Usage per step:  ex.prepare()  ->  loss.backward()  ->  ex.finish()  ->  optimizer.step().
    With gradient accumulation over k micro-batches (Lightning's `accumulate_grad_batches`, which wraps the first k - 1 backward passes in
    DDP's no_sync(); configs/multi-prompt_benchmark/asd_mv_triplane_transformer_10k.yaml:129):
        ex.prepare(sync=k == 1) -> backward -> [ex.prepare(zero=False, sync=last) -> backward] * (k - 1) -> ex.finish() -> optimizer.step()
    — the gradients are summed in place in the persistent buckets / large `.grad`s and only the LAST backward launches collectives."""

    def __init__(self, params, bucket_bytes: int = None, in_place_bytes: int = None, own=None):
        """own: a collective object with OwnCollective's interface (launch(flat) / wait() / close()) to run the units on instead of
        torch.distributed's all_reduce — what ASD_OWN_ALLREDUCE=1 builds on GPU ranks; given explicitly by the CPU test of the hand-off order"""
        self.params = [p for p in params if p.requires_grad]
        bucket_bytes = BUCKET_BYTES if bucket_bytes is None else bucket_bytes
        in_place_bytes = IN_PLACE_BYTES if in_place_bytes is None else in_place_bytes
        self.units = []           # each: dict(params=[...], flat=Tensor|None)
        cur, size = [], 0
        for p in self.params:
            nb = p.numel() * p.element_size()
            if nb >= in_place_bytes:
                self.units.append(dict(params=[p], flat=None))
                continue
            if cur and (size + nb > bucket_bytes or cur[0].dtype != p.dtype or cur[0].device != p.device):
                self.units.append(self._flat_unit(cur))
                cur, size = [], 0
            cur.append(p)
            size += nb
        if cur:
            self.units.append(self._flat_unit(cur))
        self._unit_of, self._index_of = {}, {}
        for ui, u in enumerate(self.units):
            for p in u["params"]:
                self._unit_of[id(p)] = ui
        for i, p in enumerate(self.params):
            self._index_of[id(p)] = i
        self._check_same_units_everywhere()
        # which parameters received a gradient on ANY rank this step (one tiny MAX all-reduce next to the buckets): a parameter no
        # rank touched keeps `.grad = None`, so the optimizer skips it exactly as the single-process path does (no weight decay, no
        # moment decay, no step count for an unused branch)
        dev = self.params[0].device if self.params else torch.device("cpu")
        # two pinned flag buffers used alternately, each with the event behind its asynchronous upload: the host runs a step ahead of
        # the stream (nothing in a step synchronises), so the buffer of step i may still be waiting for its copy while the hooks of
        # step i + 1 write flags — into the other buffer; before a buffer is reused its own event is waited for
        self._touched_bufs = [torch.zeros(len(self.params), dtype=torch.int32, pin_memory=dev.type == "cuda") for _ in range(2)]
        self._touched_events = [None, None]
        self._touched_slot = 0
        self._touched_host = self._touched_bufs[0]
        self._touched_dev = torch.zeros(len(self.params), dtype=torch.int32, device=dev)
        self.order = list(range(len(self.units)))      # launch order (positions into self.units)
        self._order_learned = False
        self._seen_order: List[int] = []
        # opt-in: the library's own communicator instead of torch.distributed's (GPU, fp32 gradients only)
        self._own = own
        if (own is None and os.environ.get("ASD_OWN_ALLREDUCE", "0") == "1" and is_distributed() and dist.get_backend() == "nccl" and self.params
                and all(p.is_cuda and p.dtype == torch.float32 for p in self.params)):
            self._own = OwnCollective(self.params[0].device)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._armed, self._sync = False, True
        self.exposed_ms_events = None                   # (start, end) CUDA events of the last finish(): un-overlapped exchange time
        self.prepare_called = 0

    ALIGN_BYTES = 16          # every gradient view starts on a 16-byte boundary: the optimizer kernels use 16-byte accesses

    @classmethod
    def _flat_unit(cls, ps):
        step = max(1, cls.ALIGN_BYTES // ps[0].element_size())
        offs, off = [], 0
        for p in ps:
            offs.append(off)
            off += (p.numel() + step - 1) // step * step
        flat = torch.zeros(off, dtype=ps[0].dtype, device=ps[0].device)
        views = [flat[o:o + p.numel()].view_as(p) for o, p in zip(offs, ps)]
        return dict(params=list(ps), flat=flat, views=views)

    def _check_same_units_everywhere(self) -> None:
        """the ranks must cut the exchange into the same units (same parameter list, same sizes): a mismatch would otherwise show up
        as a hang or as silently mis-added gradients inside the first collective"""
        if not is_distributed():
            return
        sig = [len(self.units)]
        for u in self.units:
            sig += [len(u["params"]), sum(p.numel() for p in u["params"]), 0 if u["flat"] is None else u["flat"].numel()]
        dev = self.params[0].device if self.params and dist.get_backend() == "nccl" else "cpu"
        n = torch.tensor([len(sig)], dtype=torch.int64, device=dev)
        dist.broadcast(n, src=0)
        ref = torch.tensor(sig if len(sig) == int(n) else [0] * int(n), dtype=torch.int64, device=dev)
        mine = ref.clone()
        dist.broadcast(ref, src=0)
        same = torch.tensor([int(len(sig) == int(n) and torch.equal(ref, mine))], dtype=torch.int64, device=dev)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        if int(same) != 1:
            raise RuntimeError(f"GradientExchange: rank {dist.get_rank()} built a different unit table than rank 0 "
                               f"({len(self.units)} units, signature {sig[:16]}...): the ranks do not hold the same trainable parameters")

    # ---- per step -----------------------------------------------------------------------------------------------------------
    def prepare(self, zero: bool = True, sync: bool = True) -> None:
        """zero: replaces optimizer.zero_grad() — bucketed gradients are zeroed views of their bucket, large ones start undefined (the
        first backward of an optimizer step); zero = False keeps what the earlier micro-batches accumulated.  sync: the coming backward
        is the last one before the optimizer step, its hooks launch the collectives; sync = False is DDP's no_sync() — the hooks only
        record which parameters received a gradient."""
        if zero:
            for u in self.units:
                if u["flat"] is None:
                    u["params"][0].grad = None
                else:
                    u["flat"].zero_()
                    for p, v in zip(u["params"], u["views"]):
                        if p.grad is not v:
                            p.grad = v
            self._touched_slot ^= 1
            if self._touched_events[self._touched_slot] is not None:     # uploaded two steps ago: normally long complete
                self._touched_events[self._touched_slot].synchronize()
                self._touched_events[self._touched_slot] = None
            self._touched_host = self._touched_bufs[self._touched_slot]
            self._touched_host.zero_()
        self._pending = [len(u["params"]) for u in self.units]
        self._ready = [False] * len(self.units)
        self._launched = 0                              # number of positions of self.order already launched
        self._works = []
        self._seen_order = []
        self._armed, self._sync = True, bool(sync)
        self.prepare_called += 1

    def _on_grad(self, p) -> None:
        if not self._armed:
            return
        ui = self._unit_of[id(p)]
        self._touched_host[self._index_of[id(p)]] = 1
        if not self._sync:                              # an accumulation micro-batch: nothing is exchanged yet
            return
        self._pending[ui] -= 1
        if self._pending[ui] == 0:
            self._ready[ui] = True
            self._seen_order.append(ui)
            if is_distributed() and self._order_learned:
                self._launch_ready()

    def _launch_ready(self, force: bool = False) -> None:
        avg = dist.ReduceOp.AVG if dist.get_backend() == "nccl" else dist.ReduceOp.SUM
        while self._launched < len(self.order):
            ui = self.order[self._launched]
            if not (self._ready[ui] or force):
                return
            u = self.units[ui]
            if u["flat"] is None:
                p = u["params"][0]
                if p.grad is None:                      # this rank produced nothing for it: contribute zeros
                    p.grad = torch.zeros_like(p)
                buf = p.grad if p.grad.is_contiguous() else None
                if buf is None:
                    p.grad = p.grad.contiguous()
                    buf = p.grad
                if self._own is not None:
                    self._own.launch(buf.view(-1))
                else:
                    self._works.append(dist.all_reduce(buf.view(-1), op=avg, async_op=True))
            elif self._own is not None:
                self._own.launch(u["flat"])
            else:
                self._works.append(dist.all_reduce(u["flat"], op=avg, async_op=True))
            self._launched += 1

    def finish(self) -> None:
        """launch what the hooks could not (units without a gradient on this rank), wait, turn sums into means."""
        self._armed = False
        if not is_distributed():
            return
        on_gpu = self.params and self.params[0].is_cuda
        if on_gpu:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        self._launch_ready(force=True)
        self._touched_dev.copy_(self._touched_host, non_blocking=True)
        if on_gpu:
            self._touched_events[self._touched_slot] = torch.cuda.Event()
            self._touched_events[self._touched_slot].record()
        self._works.append(dist.all_reduce(self._touched_dev, op=dist.ReduceOp.MAX, async_op=True))
        for w in self._works:
            w.wait()
        if self._own is not None:
            self._own.wait()
        if dist.get_backend() != "nccl":
            world = dist.get_world_size()
            for u in self.units:
                (u["flat"] if u["flat"] is not None else u["params"][0].grad).div_(world)
        if on_gpu:
            e1.record()
            self.exposed_ms_events = (e0, e1)
        if int(self._touched_host.sum()) < len(self.params):
            # this rank produced nothing for some parameter: read the global flags (the only device->host read of the exchange, and
            # only on ranks / steps with such a parameter) and leave what NO rank touched without a gradient
            for i, hit in enumerate(self._touched_dev.tolist()):
                if not hit:
                    self.params[i].grad = None
        if not self._order_learned:                     # adopt rank 0's readiness order of this first step
            seen = self._seen_order + [i for i in range(len(self.units)) if i not in self._seen_order]
            t = torch.tensor(seen, dtype=torch.int64, device=self.params[0].device if on_gpu else "cpu")
            dist.broadcast(t, src=0)
            self.order = [int(v) for v in t.tolist()]
            self._order_learned = True

    def exposed_ms(self) -> float:
        """GPU time between the end of this rank's backward and the end of the exchange in the last step (0 if fully hidden)."""
        if self.exposed_ms_events is None:
            return 0.0
        e0, e1 = self.exposed_ms_events
        e1.synchronize()
        return e0.elapsed_time(e1)

    def close(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self._own is not None:
            self._own.close()
            self._own = None


def allreduce_mean_grads(optimizer: torch.optim.Optimizer) -> None:
    """one-shot form (no overlap): mean of every gradient over the ranks, after the backward has finished.  Every parameter with
    requires_grad takes part in a fixed order — a missing gradient contributes zeros — so the ranks always issue the same
    collectives."""
    if not is_distributed():
        return
    world = dist.get_world_size()
    params = [p for grp in optimizer.param_groups for p in grp["params"] if p.requires_grad]
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        elif not p.grad.is_contiguous():
            p.grad = p.grad.contiguous()
    avg = dist.ReduceOp.AVG if dist.get_backend() == "nccl" else dist.ReduceOp.SUM
    works, flats, cur, size = [], [], [], 0

    def flush():
        nonlocal cur, size
        if cur:
            flat = torch.cat([q.grad.reshape(-1) for q in cur])
            flats.append((flat, cur))
            works.append(dist.all_reduce(flat, op=avg, async_op=True))
        cur, size = [], 0

    for p in params:
        nb = p.numel() * p.element_size()
        if nb >= IN_PLACE_BYTES:
            works.append(dist.all_reduce(p.grad.view(-1), op=avg, async_op=True))
            continue
        if cur and (size + nb > BUCKET_BYTES or cur[0].dtype != p.dtype):
            flush()
        cur.append(p)
        size += nb
    flush()
    for w in works:
        w.wait()
    if avg == dist.ReduceOp.SUM:
        for p in params:
            if p.numel() * p.element_size() >= IN_PLACE_BYTES:
                p.grad.div_(world)
    for flat, ps in flats:
        if avg == dist.ReduceOp.SUM:
            flat.div_(world)
        off = 0
        for q in ps:
            q.grad.copy_(flat[off:off + q.numel()].view_as(q.grad))
            off += q.numel()


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """identical initial parameters and buffers on every rank (DDP broadcasts from rank 0 at wrap time).  Written through
    copy_() so that tensor version counters move and caches keyed on them (the occupancy grid's packed bits) are rebuilt."""
    if not is_distributed():
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            if t.numel() == 0:
                continue
            u = t.detach().to(torch.uint8) if t.dtype == torch.bool else t.detach().clone()   # e.g. the occupancy grid's `binaries`
            dist.broadcast(u, src=src)
            t.copy_(u.to(t.dtype))
