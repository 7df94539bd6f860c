#!/bin/bash
# rocprofv3 --kernel-trace --stats over the other BASELINE.json configs (scripts/bench_configs.py), one
# pass each; summaries land in gpurun_out/<tag>_<config>.txt.  Usage (GPU box): bash scripts/profile_configs.sh r01
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in c1 c3 c4 c5 c5s; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $out/${tag}_cfg_$c -o cfg -- python $R/scripts/bench_configs.py $c --reps 3 > $out/${tag}_cfg_$c.log 2>&1
  ( cd $R; echo "## config $c: python scripts/bench_configs.py $c --reps 3"; grep -E "whole job|kernel |TFLOP|rel err|com \(" $out/${tag}_cfg_$c.log; python scripts/rocpd_summary.py $out/${tag}_cfg_$c/cfg_results.db | grep -E "^kernel|ltmi::|hipfft|rocfft" | head -12; echo ) > $out/${tag}_cfg_$c.txt 2>&1
done
cat $out/${tag}_cfg_c*.txt
