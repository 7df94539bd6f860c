mkdir -p gpurun_out/r5a
o=gpurun_out/r5a/cryst_corr.txt; : > $o
for dt in uint16 float32 uint8; do
  echo "== $dt" >> $o
  DTYPE=$dt python scripts/bench_cryst_kernel.py 2>&1 | grep "frames " >> $o
  DTYPE=$dt CORR=1 python scripts/bench_cryst_kernel.py 2>&1 | grep "frames " >> $o
  DTYPE=$dt CORR=1 LTMI_CRYST_CORR_PASS=1 python scripts/bench_cryst_kernel.py 2>&1 | grep "frames " >> $o
  DTYPE=$dt CORR=1 LTMI_FFT_FUSED=0 python scripts/bench_cryst_kernel.py 2>&1 | grep "frames " >> $o
done
cat $o
