#!/bin/bash
# parts in turn (default from 8 parts on) against contiguous parts (LTMI_KSPLIT_STRIDED=0) on other shapes
mkdir -p gpurun_out/r5a
o=gpurun_out/r5a/small_strided_shapes.txt; : > $o
run() { for v in 0 default; do echo "== $* LTMI_KSPLIT_STRIDED=$v" | tee -a $o; if [ $v = 0 ]; then env LTMI_KSPLIT_STRIDED=0 "$@" PADS=0 timeout 300 python scripts/bench_small_stride.py 2>&1 | grep -v amdgpu.ids | tee -a $o; else env "$@" PADS=0 timeout 300 python scripts/bench_small_stride.py 2>&1 | grep -v amdgpu.ids | tee -a $o; fi; done; }
run DET=256 NMASKS=16 DTYPE=uint16 SIZES=1024,2048,4096,8192,65536
run DET=512 NMASKS=3 DTYPE=uint16 NFRAMES=16384 SIZES=256,512,1024,2048,16384
run DET=256 NMASKS=16 DTYPE=float32 NFRAMES=32768 SIZES=1024,2048,4096,32768
run DET=256 NMASKS=48 DTYPE=uint16 SIZES=1024,4096,65536
run DET=1024 NMASKS=16 DTYPE=uint16 NFRAMES=4096 SIZES=128,256,1024,4096
