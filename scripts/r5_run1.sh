set -x
mkdir -p gpurun_out/r5a
for k in all half quarter; do
  python scripts/bench_sparse.py --keep $k --only 40 > gpurun_out/r5a/bell_$k.txt 2>&1
  LTMI_SPARSE_SCATTER=1 python scripts/bench_sparse.py --keep $k --only 40 > gpurun_out/r5a/scatter_$k.txt 2>&1
done
python scripts/model_fold.py > gpurun_out/r5a/model_fold.txt 2>&1
tail -n 5 gpurun_out/r5a/*.txt
