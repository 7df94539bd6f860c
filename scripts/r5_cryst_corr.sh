# CrystallinityUDF's kernel call on RAW frames with dark + gain + 50 dead pixels: corrections inside the row stage (round 5),
# the conversion pass + fused kernel (round 4, LTMI_CRYST_CORR_PASS=1), the hipFFT route; uncorrected fused kernel beside them
mkdir -p gpurun_out/r5a
o=gpurun_out/r5a/cryst_corr.txt; : > $o
for cfg in "128 65536 32" "256 16384 64" "512 4096 128" "1024 1024 256"; do
  set -- $cfg
  echo "== ${1}x${1} uint16, $2 frames" >> $o
  SIG=$1 N=$2 RAD_OUT=$3 python scripts/bench_cryst_kernel.py 2>&1 | grep "frames " >> $o
  SIG=$1 N=$2 RAD_OUT=$3 CORR=1 python scripts/bench_cryst_kernel.py 2>&1 | grep "frames " >> $o
  SIG=$1 N=$2 RAD_OUT=$3 CORR=1 LTMI_CRYST_CORR_PASS=1 python scripts/bench_cryst_kernel.py 2>&1 | grep "frames " >> $o
  SIG=$1 N=$2 RAD_OUT=$3 CORR=1 LTMI_FFT_FUSED=0 python scripts/bench_cryst_kernel.py 2>&1 | grep "frames " >> $o
done
for dt in float32 uint8; do
  echo "== 256x256 $dt, 16384 frames" >> $o
  DTYPE=$dt CORR=1 python scripts/bench_cryst_kernel.py 2>&1 | grep "frames " >> $o
  DTYPE=$dt CORR=1 LTMI_CRYST_CORR_PASS=1 python scripts/bench_cryst_kernel.py 2>&1 | grep "frames " >> $o
done
cat $o
