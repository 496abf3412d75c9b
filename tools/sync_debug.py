"""every host<->device synchronisation inside the steady-state steps of a workload (torch.cuda.set_sync_debug_mode('warn')):
    python tools/sync_debug.py [workload]"""
import os, sys, warnings, collections, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "asd_sd_3dconv_net"
torch.cuda.set_device(0)
torch.set_num_threads(1)
cfg, system, data = bench.build_system("hip", seed=10, workload=wl)
dev = torch.device("cuda", 0)
to_device = bench.to_device
batch = to_device(data.collate(), dev)
for _ in range(6):
    system.train_one_step(batch)
    batch = to_device(data.collate(), dev)
torch.cuda.synchronize()
seen = collections.Counter()
def hook(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" in str(message):
        st = [f for f in traceback.extract_stack() if "/repo/" in f.filename and "sync_debug" not in f.filename]
        where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-3:])
        seen[where] += 1
warnings.showwarning = hook
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
for _ in range(4):
    system.train_one_step(batch)
    batch = to_device(data.collate(), dev)
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
print(f"{wl}: synchronising calls over 4 steps")
for k, v in seen.most_common():
    print(f"  {v:3d} x  {k}")
if not seen:
    print("  none")
