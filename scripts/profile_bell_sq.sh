# SQ counter passes over the sparse micro-benchmark (C4 stack, 16 384 frames), full kernel and LTMI_BELL_ABLATE=1
# (no frame copies): which queue fills when the frame copies run next to the record stream.  SQ_* only (TA_* / TCP_*
# passes hung rocprofv3 on this pool in round 1).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_IDX_ACTIVE"
P3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA"
for abl in 0 1; do
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    LTMI_BELL_ABLATE=$abl timeout 300 rocprofv3 --kernel-trace --pmc $P -d $R/gpurun_out/r03_bsq_${abl}_$i -o s -- python $R/scripts/bench_sparse.py --reps 3 > $R/gpurun_out/r03_bsq_${abl}_$i.log 2>&1
  done
done
cd $R
for abl in 0 1; do for i in 1 2 3; do
  echo "== ablate=$abl pass $i"
  python scripts/rocpd_summary.py gpurun_out/r03_bsq_${abl}_$i/s_results.db | grep "k_bell" | cut -c1-30,80-220
done; done
rm -rf gpurun_out/r03_bsq_*_[123]
