# round-6 PMC passes of `python bench.py` with a short step count (the full-length FETCH_SIZE pass hit its limit twice in round 4):
# bash tools/r6_pmc.sh OUT [counters...]  -> gpurun_out/OUT/pmc_<counter>_per_kernel.csv (+ pmc_roofline_kernel.txt)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r6_pmc}; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in ${@:-FETCH_SIZE WRITE_SIZE}; do
  rm -rf /tmp/pmc_$c
  timeout 700 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_under_pmc_$c.json 2> $O/pmc_$c.err
  echo "$c exit $?"
  python $R/tools/pmc_summary.py /tmp/pmc_$c $c > $O/pmc_${c}_per_kernel.csv
  python $R/tools/pmc_last.py /tmp/pmc_$c $c field_bwd_sample_kernel 4 >> $O/pmc_roofline_kernel.txt
  grep pp_kernel $O/pmc_${c}_per_kernel.csv
done
cat $O/pmc_roofline_kernel.txt
