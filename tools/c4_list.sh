# per-position durations of selected kernels inside a C4 step:  bash tools/c4_list.sh OUT "regex"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-c4list}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_wlb
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_wlb -o p -- python $R/bench.py --workload asd_sd_3dconv_net --steps 8 --warmup 4 --no-cpu-baseline > /dev/null 2> /tmp/wlb.err
DB=$(find /tmp/prof_wlb -name "*.db" | head -1)
for pat in absmax_kernel upsample_fwd_kernel act_bwd_kernel; do python $R/tools/db_steps.py $DB 6 --marker score_fwd_kernel --list $pat 2>&1 | tail -3 | cut -c1-700; done | tee $O/list.txt
