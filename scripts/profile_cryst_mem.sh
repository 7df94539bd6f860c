R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
P1="SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_WAIT_ANY"
P2="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
P3="TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"
P4="FETCH_SIZE"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  REPS=3 timeout 300 rocprofv3 --kernel-trace --pmc $P -d $R/gpurun_out/crp_$i -o s -- python $R/scripts/bench_cryst_kernel.py > $R/gpurun_out/crp_$i.log 2>&1
done
cd $R
(for i in 1 2 3 4; do
  echo "== pass $i"
  python scripts/rocpd_summary.py gpurun_out/crp_$i/s_results.db | grep "k_cryst_fused" | cut -c1-30,70-130
  tail -3 gpurun_out/crp_$i.log | grep -i "error\|invalid" 
done) > gpurun_out/cryst_mem.txt 2>&1
rm -rf gpurun_out/crp_[1234]
