"""Launches that leave CUs idle: per (kernel, grid, workgroup) of a rocprofv3 --kernel-trace CSV, workgroups per launch against the 256
CUs, sorted by total time of the launches with fewer than 512 workgroups.   python tools/underfilled.py <dir>"""
import csv, glob, os, sys
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
agg = defaultdict(lambda: [0, 0])
with open(f, newline="") as fh:
    for r in csv.DictReader(fh):
        g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) * max(1, int(r.get("Grid_Size_Y", 1) or 1)) * max(1, int(r.get("Grid_Size_Z", 1) or 1))
        w = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1) * max(1, int(r.get("Workgroup_Size_Y", 1) or 1)) * max(1, int(r.get("Workgroup_Size_Z", 1) or 1))
        k = (r["Kernel_Name"][:64], g // max(w, 1), w)
        agg[k][0] += 1
        agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
print(f"all kernels {tot / 1e6:.1f} ms")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if k[1] < 512 and t / tot > 0.002:
        print(f"{t / 1e6:8.2f} ms {100 * t / tot:5.2f} % {n:6d} calls {t / n / 1e3:8.1f} us avg  {k[1]:5d} workgroups x {k[2]:4d} threads  {k[0]}")
