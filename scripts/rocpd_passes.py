#!/usr/bin/env python3
"""The pass kernels (train_seg_kernel) of the last `win` ms of a rocpd trace: start, duration, queue, overlap with the previous pass, idle gap to it; and the
share of that window in which 0 / 1 / 2+ pass kernels were running.  Usage: rocpd_passes.py db [win_ms]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute("select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")).fetchall()
tend = max(r[2] for r in rows)
seg = [r for r in rows if "train_seg_kernel" in r[0] and r[1] >= tend - (win + 4) * 1e6 and r[2] <= tend - 4e6]
prev_end = None
for r in seg:
    s, e = r[1], r[2]
    print("%9.1f us  dur %6.1f  q %s  %s" % ((s - seg[0][1]) / 1e3, (e - s) / 1e3, r[3] if qcol else "-", ("overlap %6.1f" % ((prev_end - s) / 1e3)) if prev_end and s < prev_end else ("gap %6.1f" % ((s - prev_end) / 1e3) if prev_end else "")))
    prev_end = max(prev_end or 0, e)
ev = sorted([(r[1], 1) for r in seg] + [(r[2], -1) for r in seg])
t_prev, lvl, hist = ev[0][0], 0, {}
for t, d in ev:
    hist[lvl] = hist.get(lvl, 0) + (t - t_prev)
    lvl += d; t_prev = t
tot = sum(hist.values())
print("window %.2f ms, %d passes; time with k passes running: %s" % (tot / 1e6, len(seg), {k: round(v / tot, 3) for k, v in sorted(hist.items())}))
