mkdir -p gpurun_out/h4
(timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/h4/gputests.log 2>&1; echo EXIT $? >> gpurun_out/h4/gputests.log); tail -4 gpurun_out/h4/gputests.log
python bench.py > gpurun_out/h4/bench.json 2> gpurun_out/h4/bench.err; cut -c1-400 gpurun_out/h4/bench.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python $R/bench.py --no-cpu-baseline > /tmp/kt.log 2>&1
DB=$(ls /tmp/prof_kt/*.db 2>/dev/null | head -1); echo DB=$DB
ls /tmp/prof_kt | head
if [ -n "$DB" ]; then python $R/tools/db_steps.py $DB 15 --skip-last 3 > $R/gpurun_out/h4/step_breakdown.txt 2>&1; fi
for f in /tmp/prof_kt/*kernel_stats.csv; do cp $f $R/gpurun_out/h4/kernel_stats.csv; done
for c in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /tmp/pmc_$c.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pmc_$c $c > $R/gpurun_out/h4/pmc_${c}_per_kernel.csv
done
head -5 $R/gpurun_out/h4/step_breakdown.txt; head -4 $R/gpurun_out/h4/kernel_stats.csv | cut -c1-150; grep -i "field_bwd\|priv_reduce" $R/gpurun_out/h4/pmc_*_per_kernel.csv | cut -c1-200
