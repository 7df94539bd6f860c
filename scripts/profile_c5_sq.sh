# SQ counter passes over the C5 micro-benchmark (8192 frames of 1024 x 1024 float32, 25 complex masks):
# clock (GRBM_GUI_ACTIVE), matrix-pipe busy cycles, waits, LDS.  SQ_* / GRBM_* only.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_IDX_ACTIVE"
P3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  ONLY_DEFAULT=1 timeout 300 rocprofv3 --kernel-trace --pmc $P -d $R/gpurun_out/c5sq_$i -o s -- python $R/scripts/bench_c5.py > $R/gpurun_out/c5sq_$i.log 2>&1
done
cd $R
for i in 1 2 3; do
  echo "== pass $i"
  python scripts/rocpd_summary.py gpurun_out/c5sq_$i/s_results.db | grep "k_dense" | cut -c1-40,80-260
done
rm -rf gpurun_out/c5sq_[123]
