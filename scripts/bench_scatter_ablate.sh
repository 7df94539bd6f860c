#!/bin/bash
# timing-only ablations of k_scatter<uint16> on C4 (results are garbage; bench_sparse.py's check is skipped)
for a in 0 1 2 5 6 7; do
  echo "== LTMI_SCATTER_ABLATE=$a"
  LTMI_SCATTER_ABLATE=$a LTMI_BENCH_NOCHECK=1 timeout 120 python scripts/bench_sparse.py --only 40 "$@" 2>&1 | grep -A1 "as dispatched"
done
