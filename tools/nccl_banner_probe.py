"""Probe: with scaledreamer_amd.dist.stdout_to_stderr around init / teardown, does stdout carry only what Python printed?
(single rank; WORLD_SIZE is forced to 2-style code path by calling the pieces directly)"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from scaledreamer_amd import dist as asd_dist

torch.cuda.set_device(0)
d = tempfile.mkdtemp()
with asd_dist.stdout_to_stderr():
    dist.init_process_group(backend="nccl", init_method=f"file://{d}/r", rank=0, world_size=1)
    t = torch.zeros(1, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
print('{"json": "line"}', flush=True)
dist.barrier()
asd_dist.shutdown()
