"""Summarise rocprofv3 (rocpd sqlite) output: per-kernel time stats and PMC counter averages."""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    print(f"# {path}")
    rows = c.execute(
        "select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
        "from kernels group by name order by sum(end-start) desc").fetchall()
    total = sum(r[5] for r in rows) or 1
    print(f"{'kernel':90s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for name, n, avg, mn, mx, tot in rows[:25]:
        print(f"{name[:90]:90s} {n:6d} {avg/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} "
              f"{100*tot/total:6.2f}")
    try:
        pm = c.execute(
            "select k.name, p.counter_name, count(*), avg(p.value) from pmc_events p "
            "join kernels k on k.dispatch_id = p.dispatch_id "
            "group by k.name, p.counter_name order by k.name").fetchall()
    except Exception:
        try:
            ccols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
            namecol = 'kernel_name' if 'kernel_name' in ccols else 'name'
            pm = c.execute(
                f"select {namecol}, counter_name, count(*), avg(value) from counters_collection "
                f"group by {namecol}, counter_name order by {namecol}").fetchall()
        except Exception as e:
            pm = []
            print("no counters:", e, cols)
    if pm:
        print(f"\n{'kernel':70s} {'counter':32s} {'n':>5s} {'avg per dispatch':>20s}")
        for name, ctr, n, v in pm:
            print(f"{name[:70]:70s} {ctr:32s} {n:5d} {v:20.1f}")


if __name__ == '__main__':
    for p in sys.argv[1:]:
        main(p)
        print()
