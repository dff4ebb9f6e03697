#!/usr/bin/env python3
"""Per-kernel average of rocprofv3 PMC counters (rocpd sqlite databases, one per --pmc pass) -> text for profiles/."""
import json
import sqlite3
import sys

args = sys.argv[1:]
json_out = workload = instr_out = None
if "--instr-json" in args:                 # --instr-json OUT.json WORKLOAD: per-launch instruction counters for bench.py's roofline
    i = args.index("--instr-json")
    instr_out, workload = args[i + 1], args[i + 2]
    del args[i:i + 3]
if "--json" in args:                       # --json OUT.json WORKLOAD: per-launch HBM bytes for bench.py's roofline.traffic
    i = args.index("--json")
    json_out, workload = args[i + 1], args[i + 2]
    del args[i:i + 3]
rows = {}
launches = {}
for path in args:
    db = sqlite3.connect(path)
    try:
        for name, n in db.execute("select name, count(*) from kernels group by name"):
            launches[name] = n                 # dispatches of the profiled command (the same command in every pass)
    except Exception:
        pass
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

if json_out:
    kern = {}
    for (name, ctr), (n, avg, tot) in rows.items():
        short = name[5:] if name.startswith("void ") else name
        short = short.split("(")[0]
        kern.setdefault(short, {})[ctr + "_KB"] = round(avg, 1)
    for k, v in kern.items():
        v["hbm_bytes"] = int((2 * v.get("FETCH_SIZE_KB", 0.) + v.get("WRITE_SIZE_KB", 0.)) * 1024)
    note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (scripts/gpu_pmc.sh), KB per launch. hbm_bytes applies the "
            "gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128 B request for coalesced streams -> x2; "
            "confirmed here on apply/bps/make_decision/gram whose known read volume is exactly 2x the counter); WRITE_SIZE "
            "matched the known 4 GiB of gram_kernel 1:1.")
    import os, hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    d = os.path.join(root, "qampy_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(open(os.path.join(d, f), "rb").read())
    # kernel_sources_sha: bench.py only quotes these numbers while the kernel sources are the ones that were profiled
    json.dump(dict(workload=workload, note=note, kernel_sources_sha=h.hexdigest()[:16], kernels=kern), open(json_out, "w"), indent=1)

if instr_out:
    import os, hashlib
    kern = {}
    for (name, ctr), (n, avg, tot) in rows.items():
        short = name[5:] if name.startswith("void ") else name
        short = short.split("(")[0]
        nl = launches.get(name, 0)
        if nl:
            kern.setdefault(short, dict(launches=nl))[ctr] = round(tot / nl, 1)      # per launch: summed over the counter instances (XCDs / SEs)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    d = os.path.join(root, "qampy_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(open(os.path.join(d, f), "rb").read())
    note = ("rocprofv3 --kernel-trace --pmc <counters> in separate passes (scripts/gpu_pmc_valu.sh): per-launch totals over all counter "
            "instances; SQ_INSTS_VALU / SQ_WAVES / steps per chain = vector instructions per wave and step of train_seg_kernel")
    json.dump(dict(workload=workload, note=note, kernel_sources_sha=h.hexdigest()[:16], kernels=kern), open(instr_out, "w"), indent=1)
