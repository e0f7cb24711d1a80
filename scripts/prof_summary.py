"""Summarise a rocprofv3 rocpd sqlite database (kernel trace and/or PMC) as text.
usage: python scripts/prof_summary.py <results.db> [...]"""
import sqlite3
import sys


def short(name, n=70):
    name = name.replace("void ", "")
    return name if len(name) <= n else name[:n - 3] + "..."


for db in sys.argv[1:]:
    con = sqlite3.connect(db)
    cur = con.cursor()
    print("==", db)
    try:
        rows = cur.execute("select name, count(*), avg(end-start)/1e3, sum(end-start)/1e3, min(end-start)/1e3, "
                           "max(end-start)/1e3, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), "
                           "max(scratch_size), max(grid_x), max(workgroup_x) from kernels group by name order by 4 desc").fetchall()
        tot = sum(r[3] for r in rows) or 1.0
        print("%-72s %6s %12s %12s %10s %10s %6s | vgpr agpr sgpr lds scratch grid wg" %
              ("kernel", "calls", "avg_us", "total_us", "min_us", "max_us", "pct"))
        for r in rows[:12]:
            print("%-72s %6d %12.1f %12.1f %10.1f %10.1f %5.1f%% | %s %s %s %s %s %s %s" %
                  (short(r[0]), r[1], r[2], r[3], r[4], r[5], 100 * r[3] / tot, *r[6:]))
    except sqlite3.Error as e:
        print("no kernel table:", e)
    try:
        rows = cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                           "group by kernel_name, counter_name order by kernel_name").fetchall()
        last = None
        for k, c, v, n in rows:
            if "bgm" not in k and "causal" not in k and "_kernel" not in k:
                continue
            if k != last:
                print("-- counters (summed over %d dispatches):" % n, short(k))
                last = k
            print("   %-32s %.6g   (per dispatch %.6g)" % (c, v, v / n))
    except sqlite3.Error:
        pass
