"""GPU idle time between kernels from a rocprofv3 --kernel-trace CSV: total busy / idle per step window and the largest gaps
with the kernels around them.   python tools/idle_gaps.py <dir with *kernel_trace.csv> [n_last_kernels]"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
with open(f, newline="") as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12000
rows = rows[-n:]
busy, idle, gaps, end = 0, 0, [], rows[0][0]
for s, e, k in rows:
    if s > end:
        idle += s - end
        gaps.append((s - end, prev, k))
    busy += max(0, e - max(s, end))
    if e > end:
        end, prev = e, k
span = end - rows[0][0]
print(f"window {span/1e6:.2f} ms: busy {busy/1e6:.2f} ms ({100*busy/span:.1f} %), idle {idle/1e6:.2f} ms over {len(gaps)} gaps")
hist = {}
for g, a, b in gaps:
    key = "<2us" if g < 2000 else "2-5us" if g < 5000 else "5-20us" if g < 20000 else "20-100us" if g < 100000 else ">100us"
    hist[key] = hist.get(key, [0, 0]); hist[key][0] += 1; hist[key][1] += g
print({k: (v[0], round(v[1] / 1e6, 2)) for k, v in hist.items()})
for g, a, b in sorted(gaps, reverse=True)[:12]:
    print(f"{g/1e3:8.1f} us  after {a}  before {b}")
agg = {}
for g, a, b in gaps:
    agg[b] = agg.get(b, [0, 0]); agg[b][0] += 1; agg[b][1] += g
print("idle time by the kernel the GPU was waiting for:")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t/1e6:7.2f} ms  {c:5d} gaps  {k}")
