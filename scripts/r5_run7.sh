#!/bin/bash
# 1-byte pixels through the row-mirror fold: parity, then timing folded / unfolded
mkdir -p gpurun_out/r5a
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "two_byte_pixels" 2>&1 | tail -5
timeout 300 python scripts/bench_radial_u8.py 2>&1 | tee gpurun_out/r5a/radial_u8.txt
NOFOLD=1 timeout 300 python scripts/bench_radial_u8.py 2>&1 | tee -a gpurun_out/r5a/radial_u8.txt
