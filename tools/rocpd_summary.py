#!/usr/bin/env python3
"""Turn rocprofv3's rocpd sqlite output (ROCm 7.2 default) into the small CSV summaries kept
under profiles/:  kernel stats (calls, total, average, share) and, for --pmc runs, the mean
counter value per kernel.

    python tools/rocpd_summary.py stats  <results.db> <out.csv>
    python tools/rocpd_summary.py pmc    <results.db> <out.csv>
"""
import csv
import sqlite3
import sys


def short(name):
    return name if len(name) < 150 else name[:147] + "..."


def main():
    mode, db, out = sys.argv[1:4]
    c = sqlite3.connect(db)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        if mode == "stats":
            w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "percent"])
            rows = c.execute(
                "select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3 "
                "from kernels group by name order by 3 desc").fetchall()
            tot = sum(r[2] for r in rows)
            for r in rows:
                w.writerow([short(r[0]), r[1], "%.3f" % r[2], "%.3f" % r[3], "%.3f" % r[4], "%.3f" % r[5],
                            "%.2f" % (100 * r[2] / tot)])
        else:
            w.writerow(["kernel", "counter", "dispatches", "mean_value", "min_value", "max_value"])
            for r in c.execute(
                    "select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
                    "from counters_collection group by kernel_name, counter_name order by 4 desc"):
                w.writerow([short(r[0]), r[1], r[2], "%.3f" % r[3], "%.3f" % r[4], "%.3f" % r[5]])


if __name__ == "__main__":
    main()
