# one rocprofv3 --pmc pass of `python bench.py` (counter $2) -> gpurun_out/$1/pmc_<counter>_per_kernel.csv
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-prof}; c=${2:-FETCH_SIZE}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/bench.py --no-cpu-baseline > $O/bench_under_pmc_$c.json 2> $O/pmc_$c.err
echo "exit $?"; tail -3 $O/pmc_$c.err
python $R/tools/pmc_summary.py /tmp/pmc_$c $c > $O/pmc_${c}_per_kernel.csv
python $R/tools/pmc_last.py /tmp/pmc_$c $c field_bwd_sample_kernel 10
python $R/tools/pmc_last.py /tmp/pmc_$c $c asd_priv_reduce_kernel 10
grep pp_kernel $O/pmc_${c}_per_kernel.csv
