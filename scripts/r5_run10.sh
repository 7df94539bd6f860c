#!/bin/bash
# C5 second run (16 bins x 25 orders, sparse) at 8192 frames: float32 frames through k_scatter / the blocked image, uint16 frames
mkdir -p gpurun_out/r5a
o=gpurun_out/r5a/second_runs.txt
: > $o
timeout 600 python scripts/bench_second_runs.py c3r c5s 2>&1 | grep -v amdgpu.ids | tee -a $o
echo "== LTMI_SPARSE_SCATTER=0 (float32 frames on the blocked image)" | tee -a $o
LTMI_SPARSE_SCATTER=0 timeout 600 python scripts/bench_second_runs.py c5s 2>&1 | grep -v amdgpu.ids | tee -a $o
timeout 600 python scripts/bench_second_runs.py c5s_u16 2>&1 | grep -v amdgpu.ids | tee -a $o
