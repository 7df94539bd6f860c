mkdir -p gpurun_out/r5a
o=gpurun_out/r5a/bell_variants.txt; : > $o
run() { echo "== $*" >> $o; env "$@" python scripts/bench_sparse.py --only 40 2>&1 | grep -A1 "as dispatched" >> $o; }
L=libertem_amd/_lib
run A=0
run LTMI_BELL_TILES=2
run LTMI_LIB=$L/libltmi_nt.so
run LTMI_LIB=$L/libltmi_nt.so LTMI_BELL_ABLATE=5 LTMI_BENCH_NOCHECK=1
run LTMI_BELL_ABLATE=5 LTMI_BENCH_NOCHECK=1
run LTMI_LIB=$L/libltmi_nb4.so LTMI_BELL_TILES=2
run LTMI_LIB=$L/libltmi_nb5.so LTMI_BELL_TILES=2
run LTMI_LIB=$L/libltmi_nb5nt.so LTMI_BELL_TILES=2
run LTMI_LIB=$L/libltmi_nb5.so LTMI_BELL_TILES=2 LTMI_BELL_ABLATE=5 LTMI_BENCH_NOCHECK=1
run LTMI_LIB=$L/libltmi_nb5.so LTMI_BELL_TILES=2 LTMI_BELL_ABLATE=1 LTMI_BENCH_NOCHECK=1
run LTMI_BELL_TILES=2 LTMI_BELL_ABLATE=1 LTMI_BENCH_NOCHECK=1
cat $o
