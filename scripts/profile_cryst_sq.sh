# SQ counter passes over the fused crystallinity kernel (scripts/bench_cryst_kernel.py: 16 384 frames of
# 256 x 256 uint16): who is busy -- vector ALUs, LDS, waits.  SQ_* / GRBM_* only, one pass per counter set.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"
P3="SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_WAVE32_LDS SQ_WAVES"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  REPS=3 timeout 300 rocprofv3 --kernel-trace --pmc $P -d $R/gpurun_out/crsq_$i -o s -- python $R/scripts/bench_cryst_kernel.py > $R/gpurun_out/crsq_$i.log 2>&1
done
cd $R
(for i in 1 2 3; do
  echo "== pass $i"
  python scripts/rocpd_summary.py gpurun_out/crsq_$i/s_results.db | grep "k_cryst_fused" | cut -c1-30,70-130
done) > gpurun_out/cryst_sq.txt 2>&1
cat gpurun_out/cryst_sq.txt
rm -rf gpurun_out/crsq_[123]
