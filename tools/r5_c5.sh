# C5 (asd_mv_triplane): HIP transformer vs the library one on the same box, then the per-step kernel table of the HIP form
O=gpurun_out/${1:-r5_c5}; mkdir -p $O
for rep in 1 2; do for t in 1 0; do
  ASD_TRITX=$t timeout 600 python bench.py --workload asd_mv_triplane --steps 16 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ASD_TRITX=$t', d['value'], d['unit'], d['ms_per_step'], 'ms')" | tee -a $O/ab.txt
done; done
bash tools/workload_breakdown.sh ${1:-r5_c5} asd_mv_triplane 8 8 > $O/breakdown_stdout.txt 2>&1; head -45 $O/asd_mv_triplane_step_breakdown.txt | cut -c1-140
