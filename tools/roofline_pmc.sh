# FETCH_SIZE / WRITE_SIZE of the roofline kernels alone (tools/roofline_kernel_only.py), separate passes: bash tools/roofline_pmc.sh OUT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-rpmc}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rk_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/rk_$c -- python $R/tools/roofline_kernel_only.py > $O/roofline_only_$c.log 2>&1
  python $R/tools/pmc_summary.py /tmp/rk_$c $c > $O/pmc_${c}_roofline_kernels.csv
done
cat $O/pmc_FETCH_SIZE_roofline_kernels.csv $O/pmc_WRITE_SIZE_roofline_kernels.csv | cut -c1-160; grep plan $O/roofline_only_FETCH_SIZE.log
