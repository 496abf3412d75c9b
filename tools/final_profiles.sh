# rocprofv3 summaries of `python bench.py` for profiles/: kernel stats CSV, per-kernel PMC FETCH / WRITE (separate passes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-prof}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- python $R/bench.py --no-cpu-baseline > $O/bench_under_trace.json 2> /tmp/kt.err
for f in $(find /tmp/prof_kt -name "*kernel_stats.csv"); do cp $f $O/kernel_stats.csv; done
for c in WRITE_SIZE FETCH_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/bench.py --no-cpu-baseline > $O/bench_under_pmc_$c.json 2> /tmp/pmc_$c.err
  python $R/tools/pmc_summary.py /tmp/pmc_$c $c > $O/pmc_${c}_per_kernel.csv
  python $R/tools/pmc_last.py /tmp/pmc_$c $c field_bwd_sample_kernel 10 >> $O/pmc_roofline_kernel.txt
  python $R/tools/pmc_last.py /tmp/pmc_$c $c asd_priv_reduce_kernel 10 >> $O/pmc_roofline_kernel.txt
done
rm -f $O/pmc_roofline_kernel.txt.tmp; cat $O/pmc_roofline_kernel.txt; grep pp_kernel $O/pmc_*_per_kernel.csv; head -6 $O/kernel_stats.csv | cut -c1-140
