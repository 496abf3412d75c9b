# round-6 evidence from ONE box and one build: GPU tests, the default bench line, rocprofv3 kernel stats of `python bench.py`, the per-step
# kernel table, FETCH_SIZE / WRITE_SIZE passes (short-step), matrix-pipe counters, the secondary workloads' raw lines, the C5 step table
O=gpurun_out/${1:-r6_final}; mkdir -p $O
R=$PWD
(timeout 1200 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo EXIT $? >> $O/gputests.log); tail -3 $O/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i smoke
python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- python $R/bench.py --no-cpu-baseline --steps 60 --warmup 10 > $R/$O/bench_under_trace.json 2> /tmp/kt.err; for f in $(find /tmp/prof_kt -name "*kernel_stats.csv"); do cp $f $R/$O/kernel_stats.csv; done)
head -5 $O/kernel_stats.csv | cut -c1-150
bash tools/step_breakdown.sh ${1:-r6_final}_brk > /dev/null 2>&1; head -3 gpurun_out/${1:-r6_final}_brk/step_breakdown.txt
bash tools/r6_pmc.sh ${1:-r6_final}_pmc FETCH_SIZE WRITE_SIZE > $O/pmc_stdout.txt 2>&1; tail -4 $O/pmc_stdout.txt
bash tools/mfma_pmc.sh ${1:-r6_final}_mfma > $O/mfma_stdout.txt 2>&1; tail -3 $O/mfma_stdout.txt
: > $O/secondary_bench_lines.jsonl
for w in asd_mv_nerf asd_sd_hyper_ingp asd_sd_3dconv_net asd_mv_triplane; do timeout 600 python bench.py --workload $w --steps 16 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/secondary_bench_lines.jsonl; done
timeout 600 python bench.py --workload asd_mv_triplane --render 256 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/secondary_bench_lines.jsonl
python - <<PY
import json
for ln in open("$O/secondary_bench_lines.jsonl"):
    d = json.loads(ln); print(d["metric"], d["value"], d["ms_per_step"], d["config"].get("render", ""))
PY
bash tools/workload_breakdown.sh ${1:-r6_final}_c5 asd_mv_triplane 8 8 > /dev/null 2>&1; head -3 gpurun_out/${1:-r6_final}_c5/asd_mv_triplane_step_breakdown.txt
bash tools/workload_breakdown.sh ${1:-r6_final}_c4 asd_sd_3dconv_net 8 8 > /dev/null 2>&1; head -3 gpurun_out/${1:-r6_final}_c4/asd_sd_3dconv_net_step_breakdown.txt
bash tools/tritx_pmc.sh ${1:-r6_final}_tritx > /dev/null 2>&1; head -12 gpurun_out/${1:-r6_final}_tritx/tritx_sq_counters.txt | cut -c1-160
python tools/r6_ws_time.py $O/ws_conv_8x8_time.txt > /dev/null 2>&1; tail -4 $O/ws_conv_8x8_time.txt | cut -c1-200
(timeout 900 python tools/gemm_shapes.py > $O/gemm_shapes_stdout.txt 2> $O/gemm_shapes.err; cp gpurun_out/gemm_shapes.txt $O/gemm_shapes_time_lost.txt); head -5 $O/gemm_shapes_time_lost.txt
bash tools/r6_pmc_shapes.sh ${1:-r6_final}_shapes > /dev/null 2>&1; cat gpurun_out/${1:-r6_final}_shapes/pmc_shapes.json
(timeout 300 python tools/lib_gemm_compare.py 30 > /dev/null 2>&1; cp gpurun_out/lib_gemm_compare.txt $O/vendor_library_compare.txt); tail -1 $O/vendor_library_compare.txt
