#!/bin/bash
# small tiles: parts of a pixel-split launch take runs of 2^(v-1) mask slots in turn (LTMI_KSPLIT_STRIDED=v) instead of a contiguous range each
mkdir -p gpurun_out/r5a
for v in 0 1 2 3 4 5; do
  echo "== LTMI_KSPLIT_STRIDED=$v" | tee -a gpurun_out/r5a/small_strided.txt
  LTMI_KSPLIT_STRIDED=$v PADS=0 timeout 300 python scripts/bench_small_stride.py 2>&1 | grep -v amdgpu.ids | grep -v 65536 | tee -a gpurun_out/r5a/small_strided.txt
done
LTMI_KSPLIT_STRIDED=3 timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "lds_dma or x16 or dense" 2>&1 | tail -3
