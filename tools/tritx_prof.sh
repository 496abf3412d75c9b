# per-kernel durations of tools/tritx_time.py under rocprofv3
O=gpurun_out/${1:-tritx_prof}; mkdir -p $O
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tx_kt; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tx_kt -o kt -- python $R/tools/tritx_time.py > $R/$O/time.txt 2>&1
for f in $(find /tmp/tx_kt -name "*kernel_stats.csv"); do python - "$f" > $R/$O/kernel_stats.txt <<'PY'
import csv, sys
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i < 30: print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.1f} min {float(r["MinNs"])/1e3:8.1f} max {float(r["MaxNs"])/1e3:8.1f} pct {r["Percentage"]}')
PY
done
cat $R/$O/kernel_stats.txt
