# rocprofv3 kernel stats of a secondary workload: bash tools/workload_stats.sh OUT WORKLOAD [steps]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-wl}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_wl
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wl -o kt -- python $R/bench.py --workload $2 --steps ${3:-6} --warmup 3 --no-cpu-baseline > $O/bench_$2.json 2> /tmp/wl.err
for f in $(find /tmp/prof_wl -name "*kernel_stats.csv"); do head -45 $f > $O/$2_kernel_stats_top44.csv; done
cut -c1-150 $O/$2_kernel_stats_top44.csv | head -30; cut -c1-200 $O/bench_$2.json
