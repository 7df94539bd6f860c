#!/bin/bash
mkdir -p gpurun_out/r5a
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_udf_gpu.py -m gpu -x -q -k "banded or radial_fourier or fold" 2>&1 | grep -v amdgpu | tail -6
timeout 600 python scripts/bench_second_runs.py c5s c5s_u16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5a/second_runs_band2.txt
BINS="2 4 8" bash scripts/r5_run15.sh 2>&1 | grep "default\|banded"
