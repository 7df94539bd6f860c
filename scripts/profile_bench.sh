#!/bin/bash
# rocprofv3 passes over bench.py (C2): kernel-trace stats + separate PMC passes (never combined with
# sys/hip/hsa traces).  Usage (on the GPU box, from the repo root): bash scripts/profile_bench.sh <tag>
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $out/${tag}_stats -o bench -- $CMD > $out/${tag}_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${tag}_fetch -o bench -- $CMD > $out/${tag}_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${tag}_write -o bench -- $CMD > $out/${tag}_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $out/${tag}_sq -o bench -- $CMD > $out/${tag}_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $out/${tag}_tcc -o bench -- $CMD > $out/${tag}_tcc.log 2>&1
cd $R
for p in stats fetch write sq tcc; do
  python scripts/rocpd_summary.py $out/${tag}_$p/bench_results.db > $out/${tag}_$p.txt 2>&1
done
tail -1 $out/${tag}_stats.log
