# same-box A/B of a secondary workload over one environment switch: bash tools/r6_ab_env_workload.sh OUT VAR WORKLOAD [reps] [steps]
O=gpurun_out/${1:-r6_ab_env_w}; mkdir -p $O; V=$2; W=$3
for rep in $(seq 1 ${4:-3}); do for p in 1 0; do
  env $V=$p python bench.py --workload $W --no-cpu-baseline --no-roofline --steps ${5:-30} --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W $V=$p', d['value'], 'steps/s', d['ms_per_step'], 'ms')" | tee -a $O/ab.txt
done; done
