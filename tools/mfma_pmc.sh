# per-kernel matrix-pipe counters of a short `python bench.py` run (one --pmc pass, SQ counters only):
#   SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES ~ fraction of CU-busy time with an MFMA in flight; SQ_INSTS_VALU_MFMA_MOPS_F16 = fp16 matrix ops
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-mfma}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_VALU_MFMA_BUSY_CYCLES\|SQ_BUSY_CU_CYCLES\|SQ_INSTS_VALU_MFMA_MOPS_F16\|SQ_WAVE_CYCLES\|SQ_BUSY_CYCLES" | sort -u > $O/counters_available.txt
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d /tmp/pmc_mfma -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /tmp/mfma.json 2> /tmp/mfma.err
tail -3 /tmp/mfma.err
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16; do python $R/tools/pmc_summary.py /tmp/pmc_mfma $c > $O/pmc_${c}_per_kernel.csv; done
cat $O/counters_available.txt; head -8 $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES_per_kernel.csv | cut -c1-120
# per (kernel, grid size) averages of the same pass: one row per shape of a kernel
python $R/tools/pmc_table.py /tmp/pmc_mfma > $O/pmc_mfma_by_kernel_and_grid.txt
