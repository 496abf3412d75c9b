# per-step kernel table of asd_mv_triplane at the 256 x 256 render: bash tools/c5_256_breakdown.sh OUT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-c5_256}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c5256
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_c5256 -o p -- python $R/bench.py --workload asd_mv_triplane --render 256 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench.json 2> /tmp/c5256.err
DB=$(find /tmp/prof_c5256 -name "*.db" | head -1)
python $R/tools/db_steps.py $DB 3 --marker score_fwd_kernel > $O/step_breakdown.txt 2>&1
head -40 $O/step_breakdown.txt | cut -c1-130
