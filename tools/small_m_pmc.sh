# SQ / TCC counters of one small-M conv shape under chosen plans: bash tools/small_m_pmc.sh OUT B hw cin cout cfg:split ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d /tmp/sm_pmc1 -- python $R/tools/small_m_one.py "$@" > /tmp/sm1.log 2>&1
echo "pass1 rc $?"
timeout 150 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d /tmp/sm_pmc2 -- python $R/tools/small_m_one.py "$@" > /tmp/sm2.log 2>&1
echo "pass2 rc $?"
tail -2 /tmp/sm1.log /tmp/sm2.log
python $R/tools/pmc_table.py /tmp/sm_pmc1 > $O/sq_pass1.txt; python $R/tools/pmc_table.py /tmp/sm_pmc2 > $O/tcc_pass2.txt
cat $O/sq_pass1.txt $O/tcc_pass2.txt | cut -c1-400
