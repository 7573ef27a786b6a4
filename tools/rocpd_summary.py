#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2 default 'rocpd' SQLite output) run as text for profiles/.

  python tools/rocpd_summary.py stats  <results.db>            per-kernel time table
  python tools/rocpd_summary.py pmc    <results.db> [filter]   per-kernel counter averages
"""
import sqlite3
import sys


def stats(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("%-100s %7s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for n, k, s, a, mn, mx in rows:
        print("%-100s %7d %14.1f %12.2f %12.2f %12.2f %6.2f%%" % (n[:100], k, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    r = c.execute("select kernels.name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, workgroup_x, grid_x, grid_y, grid_z "
                  "from kernels group by name").fetchall()
    print("\n%-100s %5s %5s %5s %8s %5s %s" % ("kernel", "vgpr", "agpr", "sgpr", "lds", "wg", "grid(threads)"))
    for x in r:
        print("%-100s %5s %5s %5s %8s %5s %sx%sx%s" % ((x[0][:100],) + tuple(x[1:])))


def pmc(db, flt=""):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) "
                     "from counters_collection where kernel_name like ? group by kernel_name, counter_name "
                     "order by avg(value)*count(*) desc", ("%" + flt + "%",)).fetchall()
    print("%-90s %-14s %7s %16s %16s %16s %12s" % ("kernel", "counter", "calls", "avg", "min", "max", "avg_dur_us"))
    for n, cn, k, a, mn, mx, d in rows:
        print("%-90s %-14s %7d %16.1f %16.1f %16.1f %12.2f" % (n[:90], cn, k, a, mn, mx, d / 1e3))


def timeline(db, anchor="k_decoder", which="-1"):
    """Every kernel from `before_us` before the start of the `which`-th (0-based, negative from the end) launch of a kernel whose
    name contains `anchor` until the next such launch: start offset, duration, queue, grid, name -- who ran next to whom."""
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = c.execute("select start, end, name, %s, grid_x, workgroup_x from kernels order by start" % q).fetchall()
    anchors = [i for i, r in enumerate(rows) if anchor in r[2]]
    if not anchors:
        print("no kernel named like", anchor, "-- columns:", cols)
        return
    i0 = anchors[int(which)]
    nxt = [i for i in anchors if i > i0]
    t0 = rows[i0][0]
    lo = t0 - 2500000
    hi = rows[nxt[0]][0] - 2500000 if nxt else rows[-1][1]
    print("%10s %10s %6s %8s  %s" % ("start_us", "dur_us", "queue", "wgs", "kernel"))
    for st, en, name, qu, gx, wx in rows:
        if st < lo or st > hi:
            continue
        short = name.replace("(anonymous namespace)::", "").replace("facppg::", "").replace("void ", "").split("(")[0][:70]
        print("%10.1f %10.1f %6s %8d  %s" % ((st - t0) / 1e3, (en - st) / 1e3, qu, gx // max(wx, 1), short))


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc, "timeline": timeline}[sys.argv[1]](*sys.argv[2:])
