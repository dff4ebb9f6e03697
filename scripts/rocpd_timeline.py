#!/usr/bin/env python3
"""Timeline of the LAST pipeline step in a rocprofv3 rocpd database: every kernel from the last `pit_setup_kernel` pair (or the
given start-kernel substring) on, with start offset, duration and the idle gap before it.  Usage: rocpd_timeline.py db [start-substring] [nth-from-last]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
key = sys.argv[2] if len(sys.argv) > 2 else "pit_setup_kernel"
nth = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = db.execute("select name, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if key in r[0]]
i0 = marks[-nth] if len(marks) >= nth else 0
t0 = rows[i0][1]
prev = t0
busy = 0
for n, s, e in rows[i0:]:
    print("%9.1f us  +%7.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n[:110]))
    prev = max(prev, e)
    busy += e - s
print("span %.1f us, kernel time %.1f us" % ((prev - t0) / 1e3, busy / 1e3))
