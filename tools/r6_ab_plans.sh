# same-box A/B of the headline step: this tree's gemm_plans.json against another plan file: bash tools/r6_ab_plans.sh OUT OTHER.json [reps] [steps]
O=gpurun_out/${1:-r6_ab_plans}; mkdir -p $O; F=$2
for rep in $(seq 1 ${3:-3}); do for p in new old; do
  if [ $p = old ]; then export ASD_GEMM_PLAN_FILE=$PWD/$F; else unset ASD_GEMM_PLAN_FILE; fi
  python bench.py --no-cpu-baseline --no-roofline --steps ${4:-60} --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plans=$p', d['value'], 'steps/s', d['ms_per_step'], 'ms')" | tee -a $O/ab.txt
done; done
