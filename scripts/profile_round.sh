#!/bin/bash
# Per-config rocprofv3 passes over bench.py (C2..C5): kernel-trace stats + SEPARATE PMC passes (never
# combined with sys/hip/hsa traces), then profiles/traffic.json + a text summary for profiles/.
# Usage (on the GPU box, from the repo root):  bash scripts/profile_round.sh <tag> [configs...]
tag=${1:-r02}
shift
cfgs=${@:-c2 c3 c4 c5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in $cfgs; do
  CMD="python $R/bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
  timeout 600 rocprofv3 --kernel-trace --stats -d $out/${c}_stats -o bench -- $CMD > $out/${c}_stats.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${c}_fetch -o bench -- $CMD > $out/${c}_fetch.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${c}_write -o bench -- $CMD > $out/${c}_write.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU -d $out/${c}_sq -o bench -- $CMD > $out/${c}_sq.log 2>&1
done
cd $R
python scripts/traffic_from_rocprof.py $tag $out $cfgs > $out/summary.txt 2>&1
cp $out/summary.txt $R/gpurun_out/${tag}_configs_rocprof.txt
cp $R/profiles/traffic.json $R/gpurun_out/${tag}_traffic.json 2>/dev/null
# the rocpd databases are large (gpurun copies back <= 64 MiB): keep logs + summaries only
for c in $cfgs; do rm -rf $out/${c}_stats $out/${c}_fetch $out/${c}_write $out/${c}_sq; done
tail -5 $out/summary.txt
