# timing-only ablations of field_bwd_sample_kernel on the headline step's samples: what is left inside it after the paged scatter
O=gpurun_out/${1:-r5_field_abl}; mkdir -p $O
python tools/field_bwd_ab.py dump > $O/dump.txt 2>&1
R=$PWD; cd /tmp && export TMPDIR=/tmp
for v in ${ABL_VARIANTS:-default nocoarse now2 nocoarse_now2}; do
  lib=""; [ $v != default ] && lib=$R/scaledreamer_amd/variants/libasd_hip_$v.so
  rm -rf /tmp/abl_kt; ASD_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_kt -o kt -- python $R/tools/field_bwd_ab.py time > /dev/null 2>&1
  for f in $(find /tmp/abl_kt -name "*kernel_stats.csv"); do python -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'field_bwd_sample_kernel' in r['Name'] or 'pg_' in r['Name']: print('%-16s %-28s avg %8.1f us' % ('$v', r['Name'][:28], float(r['AverageNs'])/1e3))
" | tee -a $R/$O/abl.txt; done
done
