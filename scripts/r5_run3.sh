mkdir -p gpurun_out/r5a
o=gpurun_out/r5a/bell_ablate6.txt; : > $o
run() { echo "== $*" >> $o; env "$@" python scripts/bench_sparse.py --only 40 2>&1 | grep -A1 "as dispatched" >> $o; }
run A=0
run LTMI_BELL_ABLATE=6 LTMI_BENCH_NOCHECK=1
run LTMI_BELL_ABLATE=4 LTMI_BENCH_NOCHECK=1
run LTMI_BELL_ABLATE=5 LTMI_BENCH_NOCHECK=1
run LTMI_BELL_ABLATE=1 LTMI_BENCH_NOCHECK=1
cat $o
