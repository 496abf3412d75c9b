# bash tools/pp_interleave.sh : conv3x3_pp_kernel<4,4> duration alone vs behind a GroupNorm apply (rocprofv3 kernel stats)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for m in alone gn gn_fresh; do
  rm -rf /tmp/ppi_$m
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ppi_$m -o r -- python $R/tools/pp_interleave.py $m > /dev/null 2>&1
  echo "== $m"; for f in $(find /tmp/ppi_$m -name "*kernel_stats.csv"); do grep -E "pp_kernel|gn_apply|gn_stats" $f | cut -d, -f1-4 | cut -c1-120; done
done
