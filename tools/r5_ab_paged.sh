# same-box A/B of the headline step: paged scatter of the hash-grid gradient (default) vs the transposed-lane atomics (ASD_FIELD_PAGED=0)
O=gpurun_out/${1:-r5_ab_paged}; mkdir -p $O
for rep in 1 2 3; do for p in 1 0; do
  ASD_FIELD_PAGED=$p python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ASD_FIELD_PAGED=$p', d['value'], 'steps/s', d['ms_per_step'], 'ms  field_bwd span', d['roofline_field_bwd']['avg_launch_ms'], 'ms')" | tee -a $O/ab.txt
done; done
