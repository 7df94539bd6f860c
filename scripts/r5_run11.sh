#!/bin/bash
mkdir -p gpurun_out/r5a
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_udf_gpu.py -m gpu -x -q -k "sparse or scatter or bell or csr or rmatmul or radial" 2>&1 | tail -5
timeout 600 python scripts/bench_second_runs.py c5s 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5a/second_runs_c5s.txt
timeout 300 python scripts/bench_sparse.py --dtype float32 --only 40 2>&1 | grep -v amdgpu.ids | tail -3
