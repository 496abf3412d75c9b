import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
cfg, system, data = bench.build_system("hip", seed=10)
dev = torch.device("cuda", 0)
for _ in range(6):
    system.train_one_step(bench.to_device(data.collate(), dev))
torch.cuda.synchronize()
un = system.guidance.backend.hip_unet
key = [k for k in un._graphs if k[0] == 5][0]
g, st = un._graphs[key]
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
print("graph replay: host call ms / until done ms:", [(round(a, 2), round(b, 2)) for a, b in ts])
