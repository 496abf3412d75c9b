# idle gaps of the last steps of a short bench run: bash tools/idle_run.sh OUT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-idle}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_idle
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_idle -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/bench.json 2>/tmp/idle.err
python $R/tools/idle_gaps.py /tmp/prof_idle 40000 > $O/idle_gaps.txt; head -50 $O/idle_gaps.txt
