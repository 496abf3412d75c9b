"""Host-side blocking calls inside the last training steps of a rocprofv3 --hip-runtime-trace --kernel-trace rocpd database:
which HIP API calls (synchronize / memcpy / graph launch) the host spends its time in, and how long the GPU queue was empty.
   python tools/db_hostsync.py <p_results.db> [--skip-last S]"""
import sqlite3, sys, collections

c = sqlite3.connect(sys.argv[1])
skip = int(sys.argv[sys.argv.index("--skip-last") + 1]) if "--skip-last" in sys.argv else 3
k = c.execute("select start, end, name from kernels order by start").fetchall()
starts = [i for i, r in enumerate(k) if "march_kernel" in r[2]]
starts = [s for j, s in enumerate(starts) if j == 0 or s - starts[j - 1] > 50]
t0, t1 = k[starts[-2 - skip]][0], k[starts[-1 - skip]][0]
cols = [r[1] for r in c.execute("pragma table_info(regions)")]
rows = c.execute("select name, start, end from regions where start >= ? and start < ? order by start", (t0 - 2000000, t1)).fetchall()
agg = collections.defaultdict(lambda: [0, 0])
for n, s, e in rows:
    agg[n][0] += 1
    agg[n][1] += e - s
print(f"step window {(t1 - t0) / 1e6:.3f} ms; host API calls in it (incl. 2 ms before its first kernel):")
for n, (cnt, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {tot / 1e3:9.1f} us {cnt:5d} x  {n}")
print("blocking calls in order:")
for n, s, e in rows:
    if e - s > 30000 or "ynchronize" in n or ("Memcpy" in n and "Async" not in n):
        print(f"  t={(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  {n}")
