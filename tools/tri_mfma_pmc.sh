# SQ counters of the tri-plane field kernels (two passes of 8 SQ slots): bash tools/tri_mfma_pmc.sh OUT [args of tri_mfma_check.py]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-tfm_pmc}; mkdir -p $O; shift
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d /tmp/tp1 -- python $R/tools/tri_mfma_check.py "$@" > /tmp/tp1.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d /tmp/tp2 -- python $R/tools/tri_mfma_check.py "$@" > /tmp/tp2.log 2>&1
python $R/tools/pmc_table.py /tmp/tp1 | grep -E "tfm_|triplane_sample" > $O/sq_pass1.txt; python $R/tools/pmc_table.py /tmp/tp2 | grep -E "tfm_|triplane_sample" > $O/sq_pass2.txt
grep -E "grid +(131072|524288|262144) " $O/sq_pass1.txt $O/sq_pass2.txt | cut -c1-170
