#!/bin/bash
# radial Fourier with 2 .. 8 bins (use_sparse=True): banded image against the dense route the container took before
mkdir -p gpurun_out/r5a
o=gpurun_out/r5a/band_bins.txt; : > $o
for nb in ${BINS:-2 3 4 8}; do
  for band in 1 0; do
    echo "== n_bins=$nb LTMI_SPARSE_BAND=$([ $band = 1 ] && echo default || echo 0)" | tee -a $o
    if [ $band = 1 ]; then
      C5S_BINS=$nb C5S_FRAMES=4096 timeout 600 python scripts/bench_second_runs.py c5s 2>&1 | grep -v amdgpu.ids | tee -a $o
    else
      LTMI_SPARSE_BAND=0 C5S_BINS=$nb C5S_FRAMES=4096 timeout 600 python scripts/bench_second_runs.py c5s 2>&1 | grep -v amdgpu.ids | tee -a $o
    fi
  done
done
