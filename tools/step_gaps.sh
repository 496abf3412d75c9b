# idle gaps above a threshold (us) inside one step of `python bench.py` (kernel trace, no counters): bash tools/step_gaps.sh OUT [thr_us]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-gaps}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_gaps
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_gaps -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_trace.json 2> /tmp/gaps.err
DB=$(find /tmp/prof_gaps -name "*.db" | head -1)
python $R/tools/db_steps.py $DB 15 --skip-last 3 --gaps ${2:-8} > $O/gaps.txt 2>&1
grep -E "idle|last step" $O/gaps.txt | tail -60
