#!/usr/bin/env python3
"""From a rocprofv3 rocpd database of scripts/overlap_probe.py: the kernels of the last overlapped pass, each with start, duration and
whether a phase-search kernel was running beside it.  Usage: overlap_trace.py db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
bps = [(s, e) for n, s, e in rows if "bps_stream" in n]
marks = [i for i, r in enumerate(rows) if "pit_setup_kernel" in r[0]]
i0 = marks[-4]
t0 = rows[i0][1]
for n, s, e in rows[i0:]:
    ov = sum(max(0, min(e, be) - max(s, bs)) for bs, be in bps if "bps_stream" not in n)
    nm = n.replace("void ", "").replace("qh::", "").split("<")[0].split("(")[0][:28]
    print("%9.1f us  %8.1f us  beside-bps %6.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, ov / 1e3, nm))
