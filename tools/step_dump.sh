# program-order launch list of one headline step: bash tools/step_dump.sh OUT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-dump}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_db
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_db -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_trace.json 2> /tmp/db.err
DB=$(find /tmp/prof_db -name "*.db" | head -1)
python $R/tools/db_steps.py $DB 15 --csv $O/kernel_stats.csv --dump $O/step_launches.txt > $O/step_breakdown.txt 2>&1
head -12 $O/step_breakdown.txt
