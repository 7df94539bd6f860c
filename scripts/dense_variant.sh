#!/bin/bash
# experiment builds of the dense kernels (here, cross-compiling): scripts/dense_variant.sh <name> [-D...]
# -> libertem_amd/_lib/exp/libltmi_<name>.so with only the C5 instantiations of k_dense_lds
# (-DLTMI_DENSE_EXP); run with LTMI_LIB=<that file> python scripts/bench_c5.py
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p libertem_amd/_lib/exp
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-inline-asm \
  -fvisibility=hidden -DLTMI_DENSE_EXP "$@" -x hip -c libertem_amd/csrc/ltmi_dense.hip \
  -o libertem_amd/_lib/exp/dense_$name.o
objs=$(ls libertem_amd/_lib/obj/*.o | grep -v ltmi_dense.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o libertem_amd/_lib/exp/libltmi_$name.so \
  libertem_amd/_lib/exp/dense_$name.o $objs -L/opt/rocm/lib -lhipfft -ldl
echo built libertem_amd/_lib/exp/libltmi_$name.so
