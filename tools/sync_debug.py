"""Which Python lines synchronise the host with the GPU inside a training step: torch.cuda.set_sync_debug_mode("warn") over two steps."""
import sys, warnings, traceback
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
import bench
cfg, system, data = bench.build_system("hip", seed=10)
dev = torch.device("cuda", 0)
for _ in range(18):
    system.train_one_step(bench.to_device(data.collate(), dev))
torch.cuda.synchronize()
seen = {}
def show(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "/root/repo" in f.filename or "scaledreamer_amd" in f.filename]
    if not st:                                  # no frame of this repository on the stack: show where it comes from anyway
        st = traceback.extract_stack()[:-1]
    key = tuple((f.filename.split("/")[-1], f.lineno) for f in st[-5:])
    seen[key] = seen.get(key, 0) + 1
warnings.showwarning = show
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
for _ in range(2):
    system.train_one_step(bench.to_device(data.collate(), dev))
torch.cuda.set_sync_debug_mode("default")
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(v, "x", " <- ".join(f"{a}:{b}" for a, b in reversed(k)))
