"""Per-step kernel-family breakdown and idle time from a rocprofv3 rocpd database (the default output of rocprofv3 7.x when no
--output-format is given).   python tools/db_steps.py <p_results.db> [n_last_steps] [--csv out.csv]
A step starts at the marcher's first kernel; the LAST n steps (default 10) are averaged."""
import re, sqlite3, sys

db = sys.argv[1]
nlast = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 10
c = sqlite3.connect(db)
rows = c.execute("select start, end, name from kernels order by start").fetchall()
short = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("(anonymous namespace)::", "")).replace("at::native::", "")[:64]
# --marker NAME: the kernel that opens a step (default the marcher; the amortized workloads have none: use e.g. score_fwd_kernel,
# which runs once per step — the step boundaries are then shifted, the per-step sums are not)
marker = sys.argv[sys.argv.index("--marker") + 1] if "--marker" in sys.argv else "march_kernel"
starts = [i for i, r in enumerate(rows) if marker in r[2]]
starts = [s for k, s in enumerate(starts) if k == 0 or s - starts[k - 1] > 50]
# bench.py: the steps are followed by roofline micro-benchmarks that also march; use --skip-last S to stay inside the timed region
skip = int(sys.argv[sys.argv.index("--skip-last") + 1]) if "--skip-last" in sys.argv else 0
bracket = [i for i, r in enumerate(rows) if "asd_trace_mark_kernel" in r[2]]
if len(bracket) >= 2 and "--skip-last" not in sys.argv:
    # bench.py brackets its timed region with two empty launches: the steps are the marker launches in between, plus the closing mark
    inside = [s_ for s_ in starts if bracket[0] < s_ < bracket[1]]
    starts = inside + [bracket[1]]
    nlast = min(nlast, len(starts) - 1)
elif "--skip-last" not in sys.argv and len(starts) > nlast + 1:
    # no explicit skip: a training step is an interval between two marker launches with the typical launch count (the micro-benchmarks
    # after the timed region march with a handful of launches in between) — take the last run of nlast such intervals
    counts = [starts[i + 1] - starts[i] for i in range(len(starts) - 1)]
    med = sorted(counts)[len(counts) // 2]
    good = [abs(cn - med) <= 0.03 * med for cn in counts]
    end = None
    for i in range(len(counts), nlast - 1, -1):
        if all(good[i - nlast:i]):
            end = i
            break
    if end is not None:
        skip = len(starts) - 1 - end
a, b = starts[-nlast - 1 - skip], starts[-1 - skip]
n = nlast
seg = rows[a:b]
wall = (rows[b][0] - rows[a][0]) / n / 1e6
fam, cnt = {}, {}
busy_end, idle = seg[0][0], 0
for s, e, nm in seg:
    k = short(nm)
    fam[k] = fam.get(k, 0) + (e - s)
    cnt[k] = cnt.get(k, 0) + 1
    if s > busy_end:
        idle += s - busy_end
    busy_end = max(busy_end, e)
tot = sum(fam.values())
print(f"{n} steps: wall {wall:.3f} ms/step, kernel sum {tot / n / 1e6:.3f} ms/step, idle {idle / n / 1e6:.3f} ms/step, {len(seg) / n:.0f} launches/step")
for k, v in sorted(fam.items(), key=lambda x: -x[1])[:50]:
    print(f"{v / n / 1e6:7.3f} ms {cnt[k] / n:7.1f} x {v / cnt[k] / 1e3:8.1f} us  {k}")
if "--csv" in sys.argv:
    out = sys.argv[sys.argv.index("--csv") + 1]
    allfam = {}
    for s, e, nm in rows:
        d = allfam.setdefault(nm, [0, 0, 10**18, 0])
        d[0] += 1; d[1] += e - s; d[2] = min(d[2], e - s); d[3] = max(d[3], e - s)
    gt = sum(d[1] for d in allfam.values())
    with open(out, "w") as fh:
        fh.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
        for nm, d in sorted(allfam.items(), key=lambda x: -x[1][1])[:80]:
            fh.write(f'"{nm[:160]}",{d[0]},{d[1]},{d[1] / d[0]:.1f},{100 * d[1] / gt:.2f},{d[2]},{d[3]}\n')

if "--gaps" in sys.argv:     # idle gaps above a threshold (us) inside the last analysed step, in program order
    thr = float(sys.argv[sys.argv.index("--gaps") + 1]) * 1e3
    step = rows[starts[-2 - skip]:starts[-1 - skip]]
    end, tot_idle = step[0][0], 0
    print(f"last step: {len(step)} kernels, {(step[-1][1] - step[0][0]) / 1e6:.3f} ms")
    for i, (s_, e_, nm) in enumerate(step):
        if s_ - end > thr:
            print(f"  idle {(s_ - end) / 1e3:7.1f} us | prev: {short(step[i - 1][2])[:48]} | next: {short(nm)[:48]}   [#{i}]")
        if s_ > end:
            tot_idle += s_ - end
        end = max(end, e_)
    print(f"idle total {tot_idle / 1e6:.3f} ms")

if "--idle-by-next" in sys.argv:     # idle time of the analysed steps by the kernel the GPU was waiting for, and the 25 longest gaps
    agg, gaps, end, prev = {}, [], seg[0][0], ""
    for s_, e_, nm in seg:
        if s_ > end:
            k = short(nm)[:70]
            agg[k] = agg.get(k, [0, 0]); agg[k][0] += 1; agg[k][1] += s_ - end
            gaps.append((s_ - end, prev, k))
        if e_ > end:
            end, prev = e_, short(nm)[:50]
    print(f"idle per step by the kernel that ended the gap ({n} steps):")
    for k, (cn, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
        print(f"  {t / n / 1e3:8.1f} us/step  {cn / n:6.1f} gaps/step  {k}")
    print("longest gaps:")
    for g, a_, b_ in sorted(gaps, reverse=True)[:25]:
        print(f"  {g / 1e3:8.1f} us  after {a_}  before {b_}")

if "--list" in sys.argv:     # durations (us) of the launches of one kernel name inside the last analysed steps, in program order, averaged over the steps
    pat = sys.argv[sys.argv.index("--list") + 1]      # regular expression on the kernel name
    per = []
    for k in range(nlast):
        st = rows[starts[-nlast - 1 - skip + k]:starts[-nlast - skip + k]]
        per.append([(e_ - s_) / 1e3 for s_, e_, nm in st if re.search(pat, nm)])
    m = min(len(p_) for p_ in per)
    print(f"{pat}: {m} launches per step; mean us per position over {nlast} steps:")
    print(" ".join(f"{sum(p_[i] for p_ in per) / nlast:.1f}" for i in range(m)))

if "--dump" in sys.argv:     # every launch of the last analysed step in program order: offset from the step start (us), duration (us), name
    step = rows[starts[-2 - skip]:starts[-1 - skip]]
    with open(sys.argv[sys.argv.index("--dump") + 1], "w") as fh:
        for s_, e_, nm in step:
            fh.write(f"{(s_ - step[0][0]) / 1e3:10.1f} {(e_ - s_) / 1e3:8.1f}  {short(nm)[:110]}\n")
