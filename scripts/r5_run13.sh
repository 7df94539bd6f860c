#!/bin/bash
# chunk planner: more annealing steps for a 1024 x 1024 stack (16384 segments)
mkdir -p gpurun_out/r5a
for it in 400000 2500000 10000000; do
  echo "== LTMI_BELL_PLAN_ITERS=$it" | tee -a gpurun_out/r5a/bell_plan.txt
  ( time LTMI_BELL_PLAN_ITERS=$it timeout 900 python scripts/bench_second_runs.py c5s ) 2>&1 | grep -v amdgpu.ids | grep "ms \|real" | tee -a gpurun_out/r5a/bell_plan.txt
done
