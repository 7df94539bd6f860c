#!/bin/bash
mkdir -p gpurun_out/r5a
for i in $(seq 1 5); do
  AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1 LTMI_ABORT_BACKTRACE=1 timeout 1200 python -X faulthandler -m pytest tests -m gpu -x -v -s -p no:cacheprovider > gpurun_out/r5a/flake_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc $(grep -E 'passed|failed' gpurun_out/r5a/flake_$i.log | tail -1)"
  if [ $rc != 0 ]; then grep -v PASSED gpurun_out/r5a/flake_$i.log | tail -n 90 | cut -c1-220; break; fi
done
