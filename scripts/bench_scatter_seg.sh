#!/bin/bash
# k_scatter<uint16> on C4 for mixing units of 128 / 256 / 512 bytes and natural chunks, with / without barriers
for seg in 128 256 512; do
  for a in 0 6 2; do
    echo "== SEG=$seg ABLATE=$a"
    LTMI_SCATTER_SEG=$seg LTMI_SCATTER_ABLATE=$a LTMI_BENCH_NOCHECK=1 timeout 120 python scripts/bench_sparse.py --only 40 "$@" 2>&1 | grep -A1 "as dispatched" | sed 's/GFLOP.*//'
  done
done
