#!/usr/bin/env python3
"""Kernel timeline of scripts/threads_probe.py from a rocprofv3 rocpd database: last 12 ms, with the queue / stream of every kernel."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
print(cols)
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
tcol = "tid" if "tid" in cols else None
sel = "name, start, end" + (", %s" % qcol if qcol else ", 0") + (", %s" % tcol if tcol else ", 0") + (", stream_id" if "stream_id" in cols else ", 0")
rows = db.execute("select %s from kernels order by start" % sel).fetchall()
tend = [r for r in rows if "bps_stream" in r[0]][-1][1]
rows = [r for r in rows if r[1] > tend - 20e6 and r[1] < tend - 8e6]
t0 = rows[0][1]
for n, s, e, q, t, st in rows:
    nm = n.replace("void ", "").replace("qh::", "").split("<")[0].split("(")[0][:26]
    if "rocclr" in nm:
        continue
    print("%9.1f us %8.1f us  q %s tid %s st %s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, t, st, nm))
