"""Turn the per-config rocprofv3 passes of scripts/profile_round.sh into

  * profiles/traffic.json -- HBM bytes per launch of the dominant kernel of every config, keyed by
    (config, bench.py's kernel label, frames per launch); bench.py quotes `roofline.traffic` from it
    only when all three match what it just ran;
  * a text summary on stdout (per-kernel duration statistics + raw counters + the bench line).

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts wide streaming reads at half
their size, so reads are doubled (guides/MI355X_MICROARCH.md, HBM / rocprofv3 section).

    python scripts/traffic_from_rocprof.py <tag> <dir with <cfg>_{stats,fetch,write,sq}/> c2 c3 ...
"""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOMINANT = ('k_dense', 'k_bell_flat', 'k_bell_apply', 'k_sell', 'k_scatter')


def find_db(d):
    hits = sorted(glob.glob(os.path.join(d, '**', '*results.db'), recursive=True))
    return hits[0] if hits else None


def _table(c, prefix):
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    exact = [n for n in names if n == prefix]
    return (exact or [n for n in names if n.startswith(prefix)] or [None])[0]


def kernel_stats(db):
    c = sqlite3.connect(db)
    t = _table(c, 'kernels')
    rows = c.execute(f"select name, count(*), avg(end-start), min(end-start), max(end-start), "
                     f"sum(end-start) from {t} group by name order by sum(end-start) desc").fetchall()
    return rows


def counters(db):
    c = sqlite3.connect(db)
    for q in ("select k.name, p.counter_name, count(*), avg(p.value) from pmc_events p "
              "join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name",
              "select kernel_name, counter_name, count(*), avg(value) from counters_collection "
              "group by kernel_name, counter_name",
              "select name, counter_name, count(*), avg(value) from counters_collection "
              "group by name, counter_name"):
        try:
            return c.execute(q).fetchall()
        except Exception:
            continue
    return []


def dominant(rows, label=''):
    """the kernel the bench line's roofline is about: for k_dense_lds the instantiation with / without the
    float16-piece products (label ',f16' <-> last template argument `true`) -- bench.py times both in one run
    (f32_instruction) -- else the first of the dominant families by total time"""
    want_x16 = ',f16' in label if 'k_dense_lds' in label else None
    for name, n, avg, mn, mx, tot in rows:
        if any(k in name for k in DOMINANT):
            if want_x16 is not None and 'k_dense_lds<' in name:
                is_x16 = name.split('>(')[0].rstrip().endswith('true')
                if is_x16 != want_x16:
                    continue
            return name, n, avg, mn, mx
    return None


def bench_line(log):
    try:
        for ln in open(log):
            if ln.startswith('{') and '"roofline"' in ln:
                return json.loads(ln)
    except OSError:
        pass
    return None


def main():
    tag, d, cfgs = sys.argv[1], sys.argv[2], sys.argv[3:]
    entries = []
    for cfg in cfgs:
        print(f"## config {cfg}")
        line = bench_line(os.path.join(d, f'{cfg}_stats.log'))
        if line is None:
            print("   no bench line in the stats pass log")
            continue
        roof = line['roofline']
        print(f"   bench.py (stats pass): value {line['value']:.4g} frames/s, ms_per_step "
              f"{line['ms_per_step']:.3f}; roofline {json.dumps(roof)}")
        per = {}
        for p in ('stats', 'fetch', 'write', 'sq'):
            db = find_db(os.path.join(d, f'{cfg}_{p}'))
            if db is None:
                print(f"   pass {p}: no database")
                continue
            rows = kernel_stats(db)
            print(f"   pass {p}: {'kernel':88s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} "
                  f"{'max_us':>10s}")
            for name, n, avg, mn, mx, tot in rows[:4]:
                print(f"      {name[:96]:96s} {n:6d} {avg/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f}")
            dom = dominant(rows, roof.get('kernel', ''))
            per[p] = dict(dom=dom, counters=counters(db))
            for name, ctr, n, v in per[p]['counters']:
                if dom and name == dom[0]:
                    print(f"      counter {ctr:28s} n={n:4d} avg per dispatch {v:18.1f}")
        try:
            dom = per['stats']['dom']
            fetch = [v for name, ctr, n, v in per['fetch']['counters']
                     if name == per['fetch']['dom'][0] and ctr == 'FETCH_SIZE'][0]
            write = [v for name, ctr, n, v in per['write']['counters']
                     if name == per['write']['dom'][0] and ctr == 'WRITE_SIZE'][0]
        except (KeyError, IndexError, TypeError) as e:
            print(f"   no traffic entry: {e!r}")
            continue
        hbm = fetch * 1024 * 2 + write * 1024
        alg = roof['algorithmic_bytes_per_launch']
        # the kernel's average in the counter passes (kernel-trace + --pmc serialises dispatches, so the
        # kernel does not share the HBM with the result copies of the previous tile like in the --stats
        # pass and in the unprofiled run)
        pmc_avgs = [per[p_]['dom'][2] / 1e3 for p_ in ('fetch', 'write', 'sq')
                    if p_ in per and per[p_]['dom']]
        pmc_avg = sum(pmc_avgs) / len(pmc_avgs) if pmc_avgs else None
        # a second kernel that belongs to every launch of the dominant one (C4: k_bell_tail, the float32 products
        # the float16 image leaves out): its time is part of the launch bench.py's HIP events bracket
        tail_avgs = {}
        for p_ in ('stats', 'fetch', 'write', 'sq'):
            db = find_db(os.path.join(d, f'{cfg}_{p_}'))
            if db is None or 'k_bell_flat' not in dom[0]:
                continue
            for name, n, avg, mn, mx, tot in kernel_stats(db):
                if 'k_bell_tail' in name:
                    tail_avgs[p_] = avg / 1e3
        tail_stats = tail_avgs.get('stats')
        tail_pmc = [v for k, v in tail_avgs.items() if k != 'stats']
        tail_pmc = sum(tail_pmc) / len(tail_pmc) if tail_pmc else None
        if tail_stats is not None:
            print(f"   + k_bell_tail per launch: {tail_stats:.1f} us (stats pass), {tail_pmc} us (counter passes)")
        print(f"   => {dom[0][:60]}: rocprof avg {dom[2]/1e3:.1f} us vs HIP events "
              f"{roof['avg_launch_ms']*1e3:.1f} us; FETCH_SIZE {fetch:.1f} KiB x 1024 x 2 (gfx950) + "
              f"WRITE_SIZE {write:.1f} KiB x 1024 = {hbm:.4g} B = {hbm/alg:.3f} x algorithmic "
              f"({alg:.4g} B)")
        entries.append({
            "config": cfg, "kernel": roof['kernel'],
            "frames_per_launch": roof['frames_per_launch'],
            "hbm_bytes_per_launch": hbm, "fetch_size_kib": fetch, "write_size_kib": write,
            "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": hbm / alg,
            "rocprof_kernel": dom[0], "rocprof_avg_us": dom[2] / 1e3, "rocprof_calls": dom[1],
            "rocprof_pmc_pass_avg_us": pmc_avg,
            "rocprof_tail_avg_us": tail_stats, "rocprof_tail_pmc_pass_avg_us": tail_pmc,
            "hip_event_avg_us": roof['avg_launch_ms'] * 1e3,
            "source": f"profiles/{tag}_configs_rocprof.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                      f"separate passes over bench.py --config {cfg})",
        })
        f32 = line.get('roofline', {}).get('f32_instruction') or line.get('f32_instruction')
        if cfg == 'c2' and isinstance(f32, dict) and f32.get('kernel'):
            # the same run also timed the strict float32-instruction kernel: its own entry (config "c2_f32")
            try:
                lab = f32['kernel']
                d2 = {p_: dominant(kernel_stats(find_db(os.path.join(d, f'{cfg}_{p_}'))), lab)
                      for p_ in ('stats', 'fetch', 'write', 'sq')}
                fe = [v for name, ctr, n, v in per['fetch']['counters'] if name == d2['fetch'][0] and ctr == 'FETCH_SIZE'][0]
                wr = [v for name, ctr, n, v in per['write']['counters'] if name == d2['write'][0] and ctr == 'WRITE_SIZE'][0]
                pm = [d2[p_][2] / 1e3 for p_ in ('fetch', 'write', 'sq') if d2.get(p_)]
                entries.append({
                    "config": "c2_f32", "kernel": lab, "frames_per_launch": roof['frames_per_launch'],
                    "hbm_bytes_per_launch": fe * 2048 + wr * 1024, "fetch_size_kib": fe, "write_size_kib": wr,
                    "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (fe * 2048 + wr * 1024) / alg,
                    "rocprof_kernel": d2['stats'][0], "rocprof_avg_us": d2['stats'][2] / 1e3,
                    "rocprof_calls": d2['stats'][1], "rocprof_pmc_pass_avg_us": sum(pm) / len(pm) if pm else None,
                    "hip_event_avg_us": f32.get('kernel_avg_launch_ms', 0) * 1e3,
                    "source": f"profiles/{tag}_configs_rocprof.txt (the f32_instruction leg of bench.py --config c2)"})
                print(f"   => f32-instruction leg {d2['stats'][0][:60]}: rocprof avg {d2['stats'][2]/1e3:.1f} us "
                      f"(counter passes {sum(pm)/len(pm):.1f} us), traffic {(fe * 2048 + wr * 1024) / alg:.3f} x algorithmic")
            except Exception as e:                        # noqa: BLE001
                print(f"   no entry for the f32-instruction leg: {e!r}")
        print()
    path = os.path.join(ROOT, 'profiles', 'traffic.json')
    old = []
    try:
        old = json.load(open(path))['entries']
    except Exception:
        pass
    keep = [e for e in old if e.get('config') not in {x['config'] for x in entries}]
    json.dump({"tag": tag, "entries": keep + entries}, open(path, 'w'), indent=1)
    print(f"wrote {path}: {len(entries)} new entr(y/ies), {len(keep)} kept")


if __name__ == '__main__':
    main()
