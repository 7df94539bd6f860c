#!/bin/bash
# workspace pass size of the 512^2 / 1024^2 crystallinity path (plan batch = frames per pass)
mkdir -p gpurun_out/r6cryst
O=gpurun_out/r6cryst/batch.txt
: > $O
for sig in 512 1024; do
  n=$((sig == 512 ? 16384 : 4096))
  for rad in 64 128; do
    for b in 512 1024 2048 4096 8192 16384; do
      [ $b -gt $n ] && continue
      echo "== sig $sig rad_out $rad BATCH=$b" >> $O
      SIG=$sig N=$n RAD_OUT=$rad REPS=10 BATCH=$b timeout 300 python scripts/bench_cryst_kernel.py 2>&1 | grep -v amdgpu.ids >> $O
    done
  done
done
grep -v "^   " $O
