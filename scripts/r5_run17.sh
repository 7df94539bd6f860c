#!/bin/bash
timeout 1200 python -m pytest tests/test_udf_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "radial_fourier or banded" 2>&1 | grep -v amdgpu | tail -6
for nb in 4 8; do
  C5S_SPARSE=0 C5S_BINS=$nb C5S_FRAMES=4096 timeout 600 python scripts/bench_second_runs.py c5s 2>&1 | grep -v amdgpu.ids | grep "ms \|use_sparse\|check"
done
