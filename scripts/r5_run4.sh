mkdir -p gpurun_out/r5a
o=gpurun_out/r5a/bell_phase.txt; : > $o
run() { echo "== $*" >> $o; env "$@" python scripts/bench_sparse.py --only 40 --reps 20 2>&1 | grep -A1 "as dispatched" | tail -1 >> $o; }
L=libertem_amd/_lib
run A=0
run LTMI_LIB=$L/libltmi_ph0.so
run LTMI_LIB=$L/libltmi_ph32.so
run LTMI_LIB=$L/libltmi_ph64.so
run LTMI_LIB=$L/libltmi_ph127.so
run A=0
cat $o
