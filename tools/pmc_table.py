"""Per-kernel, per-shape (grid size) averages of every counter in rocprofv3 --pmc counter_collection CSVs under a directory."""
import csv, glob, os, sys
from collections import defaultdict
agg, n = defaultdict(float), defaultdict(int)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            k = (row["Kernel_Name"][:60], row.get("Grid_Size", ""), row["Counter_Name"])
            agg[k] += float(row["Counter_Value"]); n[k] += 1
for k in sorted(agg):
    print(f"{k[0]:60s} grid {k[1]:>9s} {k[2]:32s} {agg[k] / n[k]:16.1f}  (n={n[k]})")
