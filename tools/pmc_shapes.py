"""One of bench.py's single-shape roofline loops on its own, for a rocprofv3 --pmc pass (tools/r6_pmc_shapes.sh):
    python tools/pmc_shapes.py gemm|vae512|unet64        the loop
    python tools/pmc_shapes.py collect DIR OUT.json      per-shape HBM bytes per launch from DIR/<shape>_<counter>/**/counter_collection.csv
The matrix kernels of a loop (gemm_f16 / conv3x3 / splitk epilogues) are summed and divided by the main kernel's dispatches; FETCH_SIZE is doubled
(MI355X_MICROARCH.md: wide reads count at half their bytes on gfx950), both counters are in KB."""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "collect":
    root, out = sys.argv[2], {}
    for shape in ("gemm", "vae512", "unet64"):
        tot, main = 0.0, {}
        for counter, scale in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
            disp = {}
            for f in glob.glob(os.path.join(root, f"{shape}_{counter}", "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f, newline="")):
                    k = row["Kernel_Name"]
                    if row.get("Counter_Name") != counter or not any(s in k for s in ("gemm_f16", "conv3x3", "splitk")):
                        continue
                    tot += float(row["Counter_Value"]) * 1024.0 * scale
                    disp.setdefault(k, set()).add(row.get("Dispatch_Id", row.get("Correlation_Id")))
            main[counter] = max((len(v) for k, v in disp.items() if "splitk" not in k), default=0)
        n = main.get("FETCH_SIZE", 0)
        out[shape] = {"bytes_per_launch": tot / n if n and n == main.get("WRITE_SIZE") else None, "launches": n}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out))
    sys.exit(0)

import torch
import bench

torch.cuda.set_device(0)
which = sys.argv[1]
r = bench.roofline_gemm_kernel(reps=20) if which == "gemm" else bench.roofline_conv_kernel(which, reps=10)
print(r["kernel"], r["avg_launch_ms"])
