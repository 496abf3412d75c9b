# per-step kernel breakdown of asd_mv_triplane at --render 256:  bash tools/workload_breakdown256.sh OUT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-wlb256}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_wlb
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_wlb -o p -- python $R/bench.py --workload asd_mv_triplane --render 256 --steps 3 --warmup 2 --no-cpu-baseline > $O/bench.json 2> /tmp/wlb.err
DB=$(find /tmp/prof_wlb -name "*.db" | head -1)
python $R/tools/db_steps.py $DB 2 --marker score_fwd_kernel > $O/step_breakdown.txt 2>&1
head -30 $O/step_breakdown.txt | cut -c1-130
