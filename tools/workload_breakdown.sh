# per-step kernel breakdown of a secondary workload (steady state: the last N steps of a kernel trace):
#   bash tools/workload_breakdown.sh OUT WORKLOAD [steps] [warmup]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-wlb}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_wlb
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_wlb -o p -- python $R/bench.py --workload $2 --steps ${3:-8} --warmup ${4:-6} --no-cpu-baseline > $O/bench_$2.json 2> /tmp/wlb.err
DB=$(find /tmp/prof_wlb -name "*.db" | head -1)
python $R/tools/db_steps.py $DB 6 --marker score_fwd_kernel --csv $O/$2_kernel_stats.csv > $O/$2_step_breakdown.txt 2>&1
head -60 $O/$2_step_breakdown.txt | cut -c1-150
