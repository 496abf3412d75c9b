"""Per-kernel sums of one rocprofv3 --pmc counter from its counter_collection CSV.
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline
  python tools/pmc_summary.py /tmp/pmc_f FETCH_SIZE > profiles/rXX_pmc_FETCH_SIZE_per_kernel.csv
FETCH_SIZE / WRITE_SIZE are in KB (memory-side requests of the L2: Infinity-Cache hits are included); on gfx950 FETCH_SIZE counts
wide coalesced reads at half their bytes (MI355X_MICROARCH.md, HBM section): the `corrected` column doubles it."""
import csv, glob, os, sys
from collections import defaultdict

root, counter = sys.argv[1], sys.argv[2]
files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
agg, cnt = defaultdict(float), defaultdict(set)
for f in files:
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != counter:
                continue
            k = row["Kernel_Name"]
            agg[k] += float(row["Counter_Value"])
            cnt[k].add(row.get("Dispatch_Id", row.get("Correlation_Id")))
scale = 2.0 if counter == "FETCH_SIZE" else 1.0
print(f"kernel,dispatches,sum_{counter}_KB,avg_KB_per_dispatch,avg_MB_per_dispatch_corrected")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 400]:
    n = max(1, len(cnt[k]))
    print(f"\"{k[:90]}\",{n},{v:.1f},{v / n:.1f},{v / n * scale / 1024:.2f}")
