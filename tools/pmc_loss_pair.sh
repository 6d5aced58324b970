R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc/lp1 -o p -- python3 $R/tools/loss_one.py > $R/gpurun_out/pmc/lp1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc/lp2 -o p -- python3 $R/tools/loss_one.py > $R/gpurun_out/pmc/lp2.log 2>&1 || exit 1
cd $R && python3 tools/pmc_table.py pair $R/gpurun_out/pmc/lp1 $R/gpurun_out/pmc/lp2 > $R/gpurun_out/pmc/loss_pair_counters.txt; cat $R/gpurun_out/pmc/loss_pair_counters.txt
