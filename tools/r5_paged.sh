# paged scatter of the hash-grid gradient (csrc/field_paged.hip): parity tests, then same-box A/B of asd_field_bwd's scatter span on the
# headline step's samples (ASD_FIELD_PAGED=0: transposed-lane atomics), then the per-kernel split under rocprofv3
O=gpurun_out/${1:-r5_paged}; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_renderer_kernels.py tests/test_gpu_renderer_golden.py tests/test_gpu_amortized.py tests/test_gpu_memory_safety.py tests/test_gpu_full_step_oracle.py tests/test_gpu_eval_phase.py -m gpu -x -q > $O/tests.log 2>&1; echo EXIT $? >> $O/tests.log); tail -5 $O/tests.log
python tools/field_bwd_ab.py dump > $O/ab.txt 2>&1
for rep in 1 2; do for p in 1 0; do echo "ASD_FIELD_PAGED=$p" >> $O/ab.txt; ASD_FIELD_PAGED=$p python tools/field_bwd_ab.py time >> $O/ab.txt 2>&1; done; done
cat $O/ab.txt | grep -v "^ "
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pg_kt; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg_kt -o kt -- python $R/tools/field_bwd_ab.py time > /dev/null 2>&1
for f in $(find /tmp/pg_kt -name "*kernel_stats.csv"); do head -12 $f | cut -c1-150 > $R/$O/kernel_stats_head.csv; done
cat $R/$O/kernel_stats_head.csv
