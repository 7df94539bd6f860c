#!/bin/bash
# C5 fold kernel: stages of a pixel-split launch in turn (LTMI_FOLD_TURN=1) against a contiguous range per part (0)
mkdir -p gpurun_out/r6fold
O=gpurun_out/r6fold/turn.txt
: > $O
for fr in 8192 2048 512 128; do
  for t in 0 1 0 1; do
    echo "== frames $fr LTMI_FOLD_TURN=$t" >> $O
    LTMI_FOLD_TURN=$t timeout 300 python scripts/bench_fold.py --frames $fr --onepx 2048 2>&1 | grep -v amdgpu.ids | grep "^folded\|element-wise\|Error\|error" >> $O
  done
done
for t in 0 1; do
  echo "== power, LTMI_FOLD_TURN=$t" >> $O
  LTMI_FOLD_TURN=$t timeout 300 python scripts/power_probe_fold.py 2>&1 | grep -v amdgpu.ids | grep "^folded  \|^folded again" >> $O
done
cat $O
