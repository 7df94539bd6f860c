#!/bin/bash
# small tiles: per-workgroup rotation of the slot order (DL_ROT_MUL variants) against the shipped order, unpadded frames
mkdir -p gpurun_out/r5a
for v in "" _rg32 _rg16; do
  echo "== libltmi$v" | tee -a gpurun_out/r5a/small_rot.txt
  LTMI_LIB=$PWD/libertem_amd/_lib/libltmi$v.so PADS=0 timeout 300 python scripts/bench_small_stride.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r5a/small_rot.txt
done
LTMI_LIB=$PWD/libertem_amd/_lib/libltmi_rg32.so timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "lds_dma or x16 or dense" 2>&1 | tail -3
