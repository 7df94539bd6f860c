#!/bin/bash
mkdir -p gpurun_out/r5a
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "banded" 2>&1 | grep -v amdgpu | tail -15
timeout 600 python scripts/bench_second_runs.py c5s c5s_u16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5a/second_runs_band.txt
