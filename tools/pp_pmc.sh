# SQ counters of the window-convolution kernels (two passes of 8 SQ slots): bash tools/pp_pmc.sh OUTNAME "cfg,cfg,..."
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pp_pmc}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d /tmp/pp_pmc1 -- python $R/tools/pp_pmc.py ${2:-11,14,20,23} > /tmp/pp1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES --output-format csv -d /tmp/pp_pmc2 -- python $R/tools/pp_pmc.py ${2:-11,14,20,23} > /tmp/pp2.log 2>&1
tail -2 /tmp/pp1.log /tmp/pp2.log
python $R/tools/pmc_table.py /tmp/pp_pmc1 > $O/sq_pass1.txt; python $R/tools/pmc_table.py /tmp/pp_pmc2 > $O/sq_pass2.txt
grep -c . $O/sq_pass1.txt $O/sq_pass2.txt
