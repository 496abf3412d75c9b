# idle gaps above a threshold (us) inside the steady-state steps of a secondary workload: bash tools/workload_gaps.sh OUT WORKLOAD [thr_us]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-wgaps}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_wgaps
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_wgaps -o p -- python $R/bench.py --workload $2 --steps 10 --warmup 8 --no-cpu-baseline --no-roofline > $O/bench_$2.json 2> /tmp/wgaps.err
DB=$(find /tmp/prof_wgaps -name "*.db" | head -1)
python $R/tools/db_steps.py $DB 6 --marker score_fwd_kernel --idle-by-next > $O/gaps.txt 2>&1
head -70 $O/gaps.txt | cut -c1-200
