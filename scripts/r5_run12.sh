#!/bin/bash
# cycle breakdown of k_bell_apply on the 16-bin radial Fourier stack (BE_PROF build)
mkdir -p gpurun_out/r5a
for fr in 8192; do
  C5S_FRAMES=$fr LTMI_LIB=$PWD/libertem_amd/_lib/libltmi_prof.so timeout 600 python scripts/bench_second_runs.py c5s 2>&1 | grep -v amdgpu.ids | grep "BE_PROF\|ms " | tail -4 | tee -a gpurun_out/r5a/bell_prof.txt
  C5S_FRAMES=$fr LTMI_LIB=$PWD/libertem_amd/_lib/libltmi_prof.so timeout 600 python scripts/bench_second_runs.py c5s_u16 2>&1 | grep -v amdgpu.ids | grep "BE_PROF\|ms " | tail -4 | tee -a gpurun_out/r5a/bell_prof.txt
done
