R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export LTMI_SELL_NW=4
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/r01_sp_a -o s -- python $R/scripts/bench_sparse.py --reps 3 > $R/gpurun_out/r01_sp_a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM -d $R/gpurun_out/r01_sp_b -o s -- python $R/scripts/bench_sparse.py --reps 3 > $R/gpurun_out/r01_sp_b.log 2>&1
cd $R
python scripts/rocpd_summary.py gpurun_out/r01_sp_a/s_results.db | grep "k_sell" | cut -c1-24,70-140
python scripts/rocpd_summary.py gpurun_out/r01_sp_b/s_results.db | grep "k_sell" | cut -c1-24,70-140
