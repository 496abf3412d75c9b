# per-kernel times of the tri-plane field check (rocprofv3 kernel trace):  bash tools/tri_mfma_prof.sh OUT [args]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-tfmp}; mkdir -p $O; shift
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tfmp -o t -- python $R/tools/tri_mfma_check.py "$@" > $O/run.txt 2>&1
for f in $(find /tmp/tfmp -name "*kernel_stats.csv"); do cp $f $O/kernel_stats.csv; done
tail -8 $O/run.txt; head -14 $O/kernel_stats.csv | cut -c1-150
