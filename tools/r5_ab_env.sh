# same-box A/B of the headline step over one environment switch: bash tools/r5_ab_env.sh OUT VAR [reps]
O=gpurun_out/${1:-r5_ab_env}; mkdir -p $O; V=$2
for rep in $(seq 1 ${3:-3}); do for p in 1 0; do
  env $V=$p python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$p', d['value'], 'steps/s', d['ms_per_step'], 'ms  field_bwd call', d['roofline_field_bwd']['asd_field_bwd_call_ms'], 'ms')" | tee -a $O/ab.txt
done; done
