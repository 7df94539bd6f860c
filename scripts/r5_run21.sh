#!/bin/bash
# k_bell_apply with two accumulation levels for <= 2 tiles: parity, then the cost (blocked image forced)
timeout 1500 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "banded or sparse or bell or csr or rmatmul or scatter" 2>&1 | grep -v amdgpu | tail -4
LTMI_SPARSE_BAND=0 timeout 600 python scripts/bench_second_runs.py c5s c5s_u16 2>&1 | grep -v amdgpu.ids | grep "ms "
timeout 300 python scripts/bench_sparse.py --dtype int16 --only 40 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 python scripts/bench_sparse.py --dtype float32 --only 42 2>&1 | grep -v amdgpu.ids | tail -2
