# same-box A/B of the headline step between the regular library and a variant build (60 timed steps, no roofline legs): bash tools/r6_ab_lib.sh OUT VARIANT [reps]
O=gpurun_out/${1:-r6_ab_lib}; mkdir -p $O; V=$PWD/scaledreamer_amd/variants/libasd_hip_$2.so
for rep in $(seq 1 ${3:-3}); do for lib in "" $V; do
  ASD_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-roofline --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant $2' if '$lib' else 'regular', d['value'], 'steps/s', d['ms_per_step'], 'ms')" | tee -a $O/ab.txt
done; done
