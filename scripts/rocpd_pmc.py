#!/usr/bin/env python3
"""Per-kernel average of rocprofv3 PMC counters (rocpd sqlite databases, one per --pmc pass) -> text for profiles/."""
import sqlite3
import sys

rows = {}
for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    try:
        cur = db.execute("select name, counter_name, count(*), avg(counter_value), sum(counter_value) from pmc_events "
                         "group by name, counter_name")
        for name, ctr, n, avg, tot in cur:
            rows[(name, ctr)] = (n, avg, tot)
    except Exception as e:                     # schema differences between rocprofiler-sdk versions: show what exists
        print("query failed on %s: %s" % (path, e))
        for (t,) in db.execute("select name from sqlite_master where type in ('table','view') and name like '%pmc%'"):
            cols = [c[1] for c in db.execute("pragma table_info(%s)" % t)]
            print("  ", t, cols)
print("%-90s %-12s %6s %16s %18s" % ("kernel", "counter", "calls", "avg", "total"))
for (name, ctr), (n, avg, tot) in sorted(rows.items(), key=lambda kv: -kv[1][2]):
    print("%-90s %-12s %6d %16.1f %18.1f" % (name[:90], ctr, n, avg, tot))
