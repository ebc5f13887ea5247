#!/usr/bin/env python3
"""Development: start/end (us, relative) of the kernels of one sweep from a rocprofv3 --kernel-trace rocpd database."""
import sqlite3, sys
db, nth = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20
span = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # pre-filter launches per sweep (chunked pipeline: 4)
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# find the nth occurrence of the segment kernel as the sweep anchor
idx = [i for i, r in enumerate(rows) if "k_kmeans_score_h1" in r[0] or "k_kmeans_score_sp<" in r[0] and "h1" not in r[0]]
starts = [i for i, r in enumerate(rows) if "k_kmeans_top2_rs" in r[0]]
if not starts:
    starts = [i for i, r in enumerate(rows) if "k_kmeans_score_h1" in r[0]]
if not starts:
    starts = [i for i, r in enumerate(rows) if "k_kmeans_score_sp" in r[0]]
nth = min(nth * span, len(starts) - span - 1)
a = starts[nth]
b = starts[nth + span]
t0 = rows[a][1]
for r in rows[a - 3:b]:
    print("%9.1f %9.1f  %7.1f  %s" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[0][:70]))
