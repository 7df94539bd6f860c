#!/bin/bash
# A/B on one box: the build before the in-turn slot order (libltmi_old.so = commit 6f8b31a) against the current one, full C2 launches
mkdir -p gpurun_out/r5a
o=gpurun_out/r5a/ab_dense.txt; : > $o
for rep in 1 2 3; do
  for v in _old ""; do
    for f32 in 0 1; do
      echo "== rep $rep libltmi$v LTMI_DENSE_F32_INSTR=$f32" | tee -a $o
      LTMI_DENSE_F32_INSTR=$f32 LTMI_LIB=$PWD/libertem_amd/_lib/libltmi$v.so PADS=0 SIZES=65536 timeout 300 python scripts/bench_small_stride.py 2>&1 | grep -v amdgpu.ids | tee -a $o
    done
  done
done
