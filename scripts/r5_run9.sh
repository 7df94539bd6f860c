#!/bin/bash
# C4: does the power-of-two frame stride cost the sparse kernel bandwidth too?  padded rows, shipped kernel and the copies-only ablation
mkdir -p gpurun_out/r5a
for pad in 0 64 128 1024; do
  for abl in 0 5; do
    echo "== pad $pad px, LTMI_BELL_ABLATE=$abl" | tee -a gpurun_out/r5a/sparse_stride.txt
    LTMI_BELL_ABLATE=$abl LTMI_BENCH_NOCHECK=1 timeout 300 python scripts/bench_sparse.py --only 40 --pad $pad 2>&1 | grep -v amdgpu.ids | grep "median" | tee -a gpurun_out/r5a/sparse_stride.txt
  done
done
