# same-box A/B of a secondary workload over one environment switch: bash tools/r5_ab_workload.sh OUT WORKLOAD VAR [reps]
O=gpurun_out/${1:-r5_ab_wl}; mkdir -p $O; W=$2; V=$3
for rep in $(seq 1 ${4:-2}); do for p in 1 0; do
  env $V=$p python bench.py --workload $W --steps 16 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W $V=$p', d['value'], 'steps/s', d['ms_per_step'], 'ms')" | tee -a $O/ab.txt
done; done
