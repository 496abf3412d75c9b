# rocprofv3 summaries of `python bench.py` for profiles/: kernel stats CSV, per-kernel PMC FETCH / WRITE (separate passes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-prof}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- python $R/bench.py --no-cpu-baseline > $O/bench_under_trace.json 2> /tmp/kt.err
for f in $(find /tmp/prof_kt -name "*kernel_stats.csv"); do cp $f $O/kernel_stats.csv; done
for c in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/bench.py --no-cpu-baseline > $O/bench_under_pmc_$c.json 2> /tmp/pmc_$c.err
  python $R/tools/pmc_summary.py /tmp/pmc_$c $c > $O/pmc_${c}_per_kernel.csv
  python $R/tools/pmc_last.py /tmp/pmc_$c $c field_bwd_sample_kernel 20 >> $O/pmc_roofline_kernel.txt
  python $R/tools/pmc_last.py /tmp/pmc_$c $c asd_priv_reduce_kernel 20 >> $O/pmc_roofline_kernel.txt
done
cat $O/pmc_roofline_kernel.txt; head -6 $O/kernel_stats.csv | cut -c1-140
