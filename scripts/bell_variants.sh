#!/bin/bash
# experiment driver (GPU box): rebuild ltmi_bell.o with other -D settings and rerun the sparse bench
cd /root/repo
for v in "$@"; do
  echo "=== variant: $v"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-inline-asm -fvisibility=hidden $v -x hip -c libertem_amd/csrc/ltmi_bell.hip -o libertem_amd/_lib/obj/ltmi_bell.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o libertem_amd/_lib/libltmi.so libertem_amd/_lib/obj/*.o -L/opt/rocm/lib -lhipfft -ldl || exit 1
  for a in ${ABL:-0}; do
    LTMI_BELL_ABLATE=$a timeout 200 python scripts/bench_sparse.py $BARGS > /tmp/bell_variant.log 2>&1
    grep -A1 "as dispatched" /tmp/bell_variant.log | tail -1 | sed "s/^/ablate=$a /"
    grep "BE_PROF" /tmp/bell_variant.log | sed -n 5p
  done
done
