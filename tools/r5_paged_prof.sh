# per-kernel durations of asd_field_bwd on the dumped samples of the headline step (tools/field_bwd_ab.py), both scatter forms
O=gpurun_out/${1:-r5_paged_prof}; mkdir -p $O
python tools/field_bwd_ab.py dump > $O/dump.txt 2>&1
R=$PWD; cd /tmp && export TMPDIR=/tmp
for p in 1 0; do
  rm -rf /tmp/pg_kt; ASD_FIELD_PAGED=$p timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg_kt -o kt -- python $R/tools/field_bwd_ab.py time > $R/$O/time_$p.txt 2>&1
  for f in $(find /tmp/pg_kt -name "*kernel_stats.csv"); do python - "$f" > $R/$O/kernel_stats_paged$p.txt <<'PY'
import csv, sys
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i < 14: print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"])/1e3:9.1f} pct {r["Percentage"]}')
PY
  done
  cat $R/$O/kernel_stats_paged$p.txt
done
