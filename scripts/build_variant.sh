#!/bin/bash
# build libertem_amd/_lib/libltmi_<tag>.so with one source recompiled under extra -D flags (experiments; select it at
# run time with LTMI_LIB=<path>).   scripts/build_variant.sh <tag> <source.hip> [-D...]
set -e
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift 2
obj=libertem_amd/_lib/obj/$(basename ${src%.*})_$tag.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-inline-asm -fvisibility=hidden "$@" -x hip -c libertem_amd/csrc/$src -o $obj
others=$(ls libertem_amd/_lib/obj/*.o | grep -v "_[a-z0-9]*\.o$\|$(basename ${src%.*})\.o" || true)
others=$(for f in libertem_amd/_lib/obj/ltmi_*.o; do b=$(basename $f .o); case $b in ltmi_capi|ltmi_comm|ltmi_dense|ltmi_sparse|ltmi_reduce|ltmi_fft|ltmi_dense64|ltmi_bell|ltmi_mib|ltmi_split|ltmi_scatter|ltmi_cryst|ltmi_fold) [ "$b" != "$(basename ${src%.*})" ] && echo $f;; esac; done)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Wl,--version-script=libertem_amd/_lib/obj/ltmi.map -o libertem_amd/_lib/libltmi_$tag.so $obj $others -L/opt/rocm/lib -lhipfft -ldl
echo libertem_amd/_lib/libltmi_$tag.so
