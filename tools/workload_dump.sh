# program-order launch list + idle gaps of one step of a secondary workload: bash tools/workload_dump.sh OUT WORKLOAD [gap_us]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-wld}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_wld
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_wld -o p -- python $R/bench.py --workload $2 --steps 8 --warmup 6 --no-cpu-baseline > $O/bench_$2.json 2> /tmp/wld.err
DB=$(find /tmp/prof_wld -name "*.db" | head -1)
python $R/tools/db_steps.py $DB 6 --marker score_fwd_kernel --gaps ${3:-15} --dump $O/$2_step_launches.txt > $O/$2_step_breakdown.txt 2>&1
grep -E "steps:|idle|last step" $O/$2_step_breakdown.txt | head -60 | cut -c1-170
