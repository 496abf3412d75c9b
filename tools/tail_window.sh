# bash tools/tail_window.sh OUT WORKLOAD STEPS WINDOW_MS SKIP_MS
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-tw}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_tw
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tw -- python $R/bench.py --workload $2 --steps ${3:-12} --warmup 4 --no-cpu-baseline > $O/bench_$2.json 2>/tmp/tw.err
python $R/tools/tail_window.py /tmp/prof_tw ${4:-400} ${5:-60} > $O/$2_tail_window.txt; cut -c1-150 $O/$2_tail_window.txt | head -48; cut -c1-160 $O/bench_$2.json
