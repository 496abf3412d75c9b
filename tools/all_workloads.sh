# one bench line per workload:  bash tools/all_workloads.sh OUT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-allwl}; mkdir -p $O; cd $R
for w in asd_sd_nerf asd_mv_nerf asd_sd_hyper_ingp asd_sd_3dconv_net asd_mv_triplane; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['value'], d['unit'], d['ms_per_step'], 'ms')" | tee -a $O/lines.txt
done
timeout 600 python bench.py --workload asd_mv_triplane --render 256 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('asd_mv_triplane@256', d['value'], d['unit'], d['ms_per_step'], 'ms')" | tee -a $O/lines.txt
