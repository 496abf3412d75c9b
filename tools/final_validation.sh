mkdir -p gpurun_out/h6
(timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/h6/gputests.log 2>&1; echo EXIT $? >> gpurun_out/h6/gputests.log); grep -E "passed|failed|EXIT" gpurun_out/h6/gputests.log
python bench.py > gpurun_out/h6/bench.json 2> gpurun_out/h6/bench.err; cut -c1-330 gpurun_out/h6/bench.json
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for w in asd_sd_hyper_ingp asd_mv_nerf; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o kt -- python $R/bench.py --workload $w --no-cpu-baseline > $R/gpurun_out/h6/bench_$w.json 2>/dev/null
  for f in $(find /tmp/prof_$w -name "*kernel_stats.csv"); do head -41 $f > $R/gpurun_out/h6/${w}_kernel_stats_top40.csv; done
  cut -c1-200 $R/gpurun_out/h6/bench_$w.json
done
