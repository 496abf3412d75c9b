"""Kernel sequence of the LAST training step in a rocprofv3 --kernel-trace CSV, with every idle gap above a threshold:
which kernels the GPU waited for, in program order.   python tools/step_gaps.py <dir> [min_gap_us]"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
thr = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 6e3
rows = []
with open(f, newline="") as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# a step starts at the marcher's count kernel
starts = [i for i, r in enumerate(rows) if "march_kernel<true>" in r[2] or "march_kernel<false>" in r[2]]
starts = [s for k, s in enumerate(starts) if k == 0 or s - starts[k - 1] > 50]
a, b = starts[-2], starts[-1]
step = rows[a:b]
short = lambda n: n.replace("void ", "").replace("at::native::", "")[:70]
end = step[0][0]
tot_idle = 0
print(f"step: {len(step)} kernels, {(step[-1][1] - step[0][0]) / 1e6:.3f} ms")
for i, (s, e, n) in enumerate(step):
    if s - end > thr:
        print(f"  idle {(s - end) / 1e3:7.1f} us | prev: {short(step[i - 1][2])} | next: {short(n)}   [kernel #{i}]")
    if s > end:
        tot_idle += s - end
    end = max(end, e)
print(f"idle total {tot_idle / 1e6:.3f} ms")
