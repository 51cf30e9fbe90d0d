#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd databases (ROCm 7.2 writes sqlite, not CSV) into the text kept under profiles/.

usage: rocprof_summary.py [--json out.json] <results.db> [more.db ...]
Prints per-kernel call count / total / average duration (the `--stats` view) and, when the run collected PMC
counters, the per-kernel per-launch average of each counter.  FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section) -- both raw and corrected are printed.
"""
import sqlite3
import sys


def main():
    argv = sys.argv[1:]
    jpath, jout = None, {}
    if argv and argv[0] == "--json":
        jpath, argv = argv[1], argv[2:]
    for path in argv:
        db = sqlite3.connect(path)
        cur = db.cursor()
        print("== %s" % path)
        print("-- kernel stats (name | calls | total_ms | avg_ms | pct)")
        for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 12"):
            short = name if len(name) < 100 else name[:97] + "..."
            print("%-100s | %5d | %12.3f | %12.3f | %5.1f" % (short, calls, total / 1e3, avg / 1e3, pct))
        rows = list(cur.execute(
            "select kernel_name, counter_name, count(*), avg(value) from counters_collection "
            "group by kernel_name, counter_name order by kernel_name"))
        if rows:
            print("-- PMC (kernel | counter | launches | avg per launch)")
            for k, c, n, v in rows:
                if "rg::" not in k:
                    continue
                short = k if len(k) < 70 else k[:67] + "..."
                extra = ""
                if c == "FETCH_SIZE":
                    extra = "  = %.2f GB raw, %.2f GB with the gfx950 x2 correction" % (v * 1024 / 1e9, 2 * v * 1024 / 1e9)
                if c == "WRITE_SIZE":
                    extra = "  = %.2f GB (uncalibrated)" % (v * 1024 / 1e9)
                print("%-70s | %-14s | %3d | %.4g%s" % (short, c, n, v, extra))
                jout.setdefault(k, {})[c] = {"launches": n, "avg": v}
    if jpath:
        import json
        json.dump(jout, open(jpath, "w"), indent=1)


if __name__ == "__main__":
    main()
