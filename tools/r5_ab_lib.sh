# same-box A/B of the headline step (or a workload) between the regular library and a variant build: bash tools/r5_ab_lib.sh OUT VARIANT [WORKLOAD] [reps]
O=gpurun_out/${1:-r5_ab_lib}; mkdir -p $O; V=$PWD/scaledreamer_amd/variants/libasd_hip_$2.so; W=${3:-asd_sd_nerf}
for rep in $(seq 1 ${4:-3}); do for lib in "" $V; do
  ASD_HIP_LIB=$lib python bench.py --workload $W --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W', 'variant $2' if '$lib' else 'regular', d['value'], 'steps/s', d['ms_per_step'], 'ms')" | tee -a $O/ab.txt
done; done
