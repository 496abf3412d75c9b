"""Kernel-family breakdown of the LAST `ms` milliseconds before `skip_ms` from the end of a rocprofv3 --kernel-trace CSV (steady-state
steps of a workload whose first steps run library searches).   python tools/tail_window.py <dir> <ms> [skip_ms]"""
import csv, glob, os, re, sys
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
with open(f, newline="") as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
end = rows[-1][1] - int(float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 0)
beg = end - int(float(sys.argv[2]) * 1e6)
agg = defaultdict(lambda: [0, 0])
busy = 0
for s, e, k in rows:
    if s >= beg and e <= end:
        k = re.sub(r"\(.*", "", k.replace("void ", ""))[:90]
        agg[k][0] += 1; agg[k][1] += e - s; busy += e - s
print(f"window {float(sys.argv[2]):.0f} ms: kernels {busy / 1e6:.1f} ms")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{t / 1e6:8.2f} ms {100 * t / busy:5.1f} % {n:6d} x {t / n / 1e3:9.1f} us  {k}")
