# TCC (L2) counters of k_bell_flat on C4: shipped vs records from 16 hot KiB (LTMI_BELL_ABLATE=6), separate --pmc passes
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r5a/tcc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for abl in 0 6; do
  for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    tag=$(echo $grp | tr ' ' '_')
    LTMI_BELL_ABLATE=$abl LTMI_BENCH_NOCHECK=1 timeout 180 rocprofv3 --kernel-trace --pmc $grp -d $out/a${abl}_$tag -o r -- python $R/scripts/bench_sparse.py --only 40 --reps 5 > $out/a${abl}_$tag.log 2>&1
    echo "ablate=$abl $grp rc=$?"
  done
done
cd $R
python - <<'PY'
import glob, sqlite3, os
for db in sorted(glob.glob('gpurun_out/r5a/tcc/**/*results.db', recursive=True)):
    c = sqlite3.connect(db)
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    for q in ("select k.name, p.counter_name, count(*), avg(p.value) from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name",
              "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        try:
            rows = c.execute(q).fetchall(); break
        except Exception as e:
            rows = []
    for name, ctr, n, v in rows:
        if 'k_bell_flat' in name:
            print(db.split('/')[3], ctr, n, f"{v:.4g}")
PY
rm -rf $out/*/
