#!/usr/bin/env python3
"""Kernel-time summary (the `--stats` view) of a rocprofv3 rocpd sqlite database -> text for profiles/."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name "
                  "order by sum(end-start) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print("%-100s %6s %14s %14s %14s %14s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
for n, c, s, a, mn, mx in rows:
    print("%-100s %6d %14d %14.0f %14d %14d %6.2f%%" % (n[:100], c, s, a, mn, mx, 100.0 * s / tot))
