# same-box A/B of two environments on the headline step: bash tools/ab_env.sh OUT "ENV_A" "ENV_B" [rounds]
#   e.g. bash tools/ab_env.sh ab1 "ASD_GEMM_PLAN_FILE=tools/data/gemm_plans_r02.json" "" 3
O=gpurun_out/${1:-ab}; mkdir -p $O; : > $O/ab.txt
for i in $(seq 1 ${4:-3}); do
  for v in A B; do
    if [ $v = A ]; then E="$2"; else E="$3"; fi
    env $E python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['value'], d.get('gpu_ms_per_step_median', ''))" >> $O/ab.txt
  done
done
cat $O/ab.txt
