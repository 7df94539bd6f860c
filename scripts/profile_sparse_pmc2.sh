# further PMC passes over the sparse micro-benchmark: instruction fetch / branches, texture-address path
R=${GRAFT_REPO_ROOT:-/root/repo}
PAT=${1:-k_bell}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM -d $R/gpurun_out/r01_sp_d -o s -- python $R/scripts/bench_sparse.py --reps 3 > $R/gpurun_out/r01_sp_d.log 2>&1
# (a TA_*/TCP_* pass -- TA_TA_BUSY_sum, TCP_TCC_READ_REQ_LATENCY_sum ... -- hung rocprofv3 on this pool: not collected)
cd $R
for x in d; do
  python scripts/rocpd_summary.py gpurun_out/r01_sp_$x/s_results.db | grep "$PAT" | cut -c1-24,70-200
  tail -2 gpurun_out/r01_sp_$x.log | cut -c1-200
done
