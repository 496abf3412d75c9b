# per-step kernel-family breakdown of `python bench.py` (rocprofv3 kernel trace -> rocpd database -> tools/db_steps.py): bash tools/step_breakdown.sh OUT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-brk}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_db
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_db -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_trace.json 2> /tmp/db.err
DB=$(find /tmp/prof_db -name "*.db" | head -1)
python $R/tools/db_steps.py $DB 15 ${2:+--skip-last $2} --csv $O/kernel_stats.csv > $O/step_breakdown.txt 2>&1
python $R/tools/db_steps.py $DB 15 ${2:+--skip-last $2} --list "${3:-conv3x3_pp_kernel<4, 4>}" | tail -2 > $O/launch_list.txt
head -${4:-45} $O/step_breakdown.txt; cat $O/launch_list.txt
