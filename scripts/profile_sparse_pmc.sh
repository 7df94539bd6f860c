# PMC passes over the sparse micro-benchmark (GPU box): scripts/profile_sparse_pmc.sh [kernel-name-pattern]
R=${GRAFT_REPO_ROOT:-/root/repo}
PAT=${1:-k_bell}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/r01_sp_a -o s -- python $R/scripts/bench_sparse.py --reps 3 > $R/gpurun_out/r01_sp_a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM -d $R/gpurun_out/r01_sp_b -o s -- python $R/scripts/bench_sparse.py --reps 3 > $R/gpurun_out/r01_sp_b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d $R/gpurun_out/r01_sp_c -o s -- python $R/scripts/bench_sparse.py --reps 3 > $R/gpurun_out/r01_sp_c.log 2>&1
cd $R
for x in a b c; do
  python scripts/rocpd_summary.py gpurun_out/r01_sp_$x/s_results.db | grep "$PAT" | cut -c1-24,70-200
done
