# same-box A/B of the headline step over one environment switch: bash tools/r6_ab_env.sh OUT VAR [reps] [steps]
O=gpurun_out/${1:-r6_ab_env}; mkdir -p $O; V=$2
for rep in $(seq 1 ${3:-3}); do for p in 1 0; do
  env $V=$p python bench.py --no-cpu-baseline --no-roofline --steps ${4:-60} --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$p', d['value'], 'steps/s', d['ms_per_step'], 'ms')" | tee -a $O/ab.txt
done; done
