# copy the evidence of one tools/r6_final.sh pass (gpurun_out/<P>*) into profiles/r06_*: bash tools/r6_copy_profiles.sh r6_final2
P=gpurun_out/${1:-r6_final}
cp $P/kernel_stats.csv profiles/r06_kernel_stats.csv
cp $P/bench.json profiles/r06_bench.json
cp $P/secondary_bench_lines.jsonl profiles/r06_secondary_bench_lines.jsonl
tail -4 $P/gputests.log > profiles/r06_gpu_tests_tail.txt
cp ${P}_brk/step_breakdown.txt profiles/r06_step_breakdown.txt
for c in FETCH_SIZE WRITE_SIZE; do cp ${P}_pmc/pmc_${c}_per_kernel.csv profiles/r06_pmc_${c}_per_kernel.csv; done
cp ${P}_pmc/pmc_roofline_kernel.txt profiles/r06_pmc_roofline_kernel.txt
cp ${P}_mfma/pmc_SQ_VALU_MFMA_BUSY_CYCLES_per_kernel.csv profiles/r06_pmc_mfma_busy_cycles_per_kernel.csv
cp ${P}_mfma/pmc_SQ_BUSY_CU_CYCLES_per_kernel.csv profiles/r06_pmc_busy_cu_cycles_per_kernel.csv
cp ${P}_mfma/pmc_mfma_by_kernel_and_grid.txt profiles/r06_pmc_mfma_by_kernel_and_grid.txt
cp ${P}_c5/asd_mv_triplane_step_breakdown.txt profiles/r06_c5_triplane_step_breakdown.txt
cp ${P}_c4/asd_sd_3dconv_net_step_breakdown.txt profiles/r06_c4_3dconv_step_breakdown.txt
cat ${P}_tritx/tritx_sq_summary.txt ${P}_tritx/tritx_sq_counters.txt > profiles/r06_tritx_sq_counters.txt
cp $P/ws_conv_8x8_time.txt profiles/r06_ws_conv_8x8_time.txt
cp $P/gemm_shapes_time_lost.txt profiles/r06_gemm_shapes_time_lost.txt
[ -f gpurun_out/tritx_full_size_vs_float64.txt ] && cp gpurun_out/tritx_full_size_vs_float64.txt profiles/r06_tritx_full_size_vs_float64.txt
cp ${P}_shapes/pmc_shapes.json profiles/r06_pmc_shapes.json
cp $P/vendor_library_compare.txt profiles/r06_vendor_library_compare.txt
python - <<PY
import json
d = json.loads(open("${P}_pmc/bench_under_pmc_FETCH_SIZE.json").read().strip().splitlines()[-1])
json.dump({"samples_per_launch": d["roofline_field_bwd"]["samples_per_launch"], "source": "roofline_field_bwd.samples_per_launch of the bench line printed under the FETCH_SIZE pass"}, open("profiles/r06_pmc_field_span.json", "w"), indent=1)
PY
