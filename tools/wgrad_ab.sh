# rows per block of field_wgrad_kernel: whole asd_field_bwd call minus its sample kernel, from bench.py's roofline leg (same box)
for v in base wg1024 wg512 wg256 base wg512; do
  if [ $v = base ]; then L=scaledreamer_amd/libasd_hip.so; else L=scaledreamer_amd/variants/libasd_hip_$v.so; fi
  ASD_HIP_LIB=$L python bench.py --steps 8 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v', 'call', r['asd_field_bwd_call_ms'], 'sample kernel', r['avg_launch_ms'], 'rest', round(r['asd_field_bwd_call_ms']-r['avg_launch_ms'],4), 'step', d['ms_per_step'])"
done
