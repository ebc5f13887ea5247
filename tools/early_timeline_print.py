#!/usr/bin/env python3
"""Development: kernels of the LAST n_sweeps * k launches groups of a rocpd kernel trace of tools/early_timeline.py: every kernel
after the last constructor, with start / duration (us) relative to the first."""
import sqlite3
import sys

db = sys.argv[1]
n_sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# sweeps end with k_batch_post: take the last n_sweeps of them
posts = [i for i, r in enumerate(rows) if "k_batch_post" in r[0]]
first = posts[-n_sweeps - 1] + 1 if len(posts) > n_sweeps else 0
t0 = rows[first][1]
prev_end = t0
for r in rows[first:]:
    print("%9.1f  %7.1f  gap %6.1f  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[1] - prev_end) / 1e3, r[0][:90]))
    prev_end = r[2]
    if "k_batch_post" in r[0]:
        print("   ---- sweep ends at %.1f" % ((r[2] - t0) / 1e3))
