R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r5a/small; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $out/t -o r -- python $R/scripts/bench_small_tiles.py > $out/trace.log 2>&1
cd $R
python - > gpurun_out/r5a/small/breakdown.txt 2>&1 <<'PY'
import glob, sqlite3
db = sorted(glob.glob('gpurun_out/r5a/small/**/*results.db', recursive=True))[0]
c = sqlite3.connect(db)
t = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')") if r[0].startswith('kernels')][0]
rows = c.execute(f"select name, start, end from {t} order by start").fetchall()
# consecutive (dense, reduce) pairs: durations and gaps, grouped by the dense kernel's duration bucket
import collections
seq = [(n, s, e) for n, s, e in rows]
out = collections.defaultdict(list)
for i in range(1, len(seq) - 1):
    n, s, e = seq[i]
    if 'k_dense_lds' in n and 'k_reduce' in seq[i + 1][0] and 'k_reduce' in seq[i - 1][0]:
        out[round((e - s) / 1e3 / 5) * 5].append(((e - s) / 1e3, (seq[i + 1][1] - e) / 1e3, (seq[i + 1][2] - seq[i + 1][1]) / 1e3, (s - seq[i - 1][2]) / 1e3))
for k in sorted(out):
    a = out[k]
    import statistics as st
    print(f"dense ~{k} us: n={len(a)} dense {st.median(x[0] for x in a):.1f} gap->reduce {st.median(x[1] for x in a):.1f} reduce {st.median(x[2] for x in a):.1f} gap reduce->next dense {st.median(x[3] for x in a):.1f}")
PY
rm -rf $out/t
