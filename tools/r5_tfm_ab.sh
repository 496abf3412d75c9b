# the weight-gradient kernel of the tri-plane field with the shallow (regular build) / deep (variant tfmdeep) lookup pipeline: kernel times and accuracy
O=gpurun_out/${1:-r5_tfm_ab}; mkdir -p $O
for lib in "" $PWD/scaledreamer_amd/variants/libasd_hip_tfmdeep.so; do
  echo "== ${lib:-regular (shallow)}" | tee -a $O/ab.txt
  ASD_HIP_LIB=$lib timeout 300 python tools/tri_mfma_check.py 2>&1 | grep -E "backward|bwd" | tee -a $O/ab.txt
done
for rep in 1 2; do for lib in "" $PWD/scaledreamer_amd/variants/libasd_hip_tfmdeep.so; do
  ASD_HIP_LIB=$lib python bench.py --workload asd_mv_triplane --render 256 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5-256', 'deep' if '$lib' else 'shallow', d['ms_per_step'], 'ms')" | tee -a $O/ab.txt
done; done
