"""Average GPU duration per (kernel, grid size, LDS) from a rocprofv3 --kernel-trace CSV: the true per-shape launch times inside the
graph-replayed step (eager micro-benchmarks of < 12 us kernels measure the host).   python tools/trace_by_grid.py <dir> [substr]"""
import csv, glob, os, sys
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = defaultdict(lambda: [0, 0])
with open(f, newline="") as fh:
    for r in csv.DictReader(fh):
        if sub not in r["Kernel_Name"]:
            continue
        k = (r["Kernel_Name"][:70], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("LDS_Block_Size", ""))
        a = agg[k]
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{t / 1e6:9.2f} ms total {n:6d} calls {t / n / 1e3:8.1f} us avg  grid {k[1]:>8s} lds {k[2]:>7s}  {k[0]}")
