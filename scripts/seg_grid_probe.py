"""C3 (or ns): consecutive captures through one receiver for a list of (lanes per chain, segments of stage 1, segments of stage 2) - throughput, passes, pass time.
Usage: seg_grid_probe.py workload tol  lanes:S1:S2 ...   (lanes 0 = automatic, S = 0 automatic)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from qampy_amd import _lib
if os.environ.get("QAMPY_LIB"):                  # (measurements: another build of the library, e.g. -DQH_SEG_DUAL)
    _lib.LIB_PATH = os.environ["QAMPY_LIB"]
import bench
key, tol = sys.argv[1], float(sys.argv[2])
cfg = bench.WORKLOADS[key]; nsym = cfg["nsym"]
sig = bench.make_input(cfg, nsym, 1000)
K = 24
for spec in sys.argv[3:]:
    lanes, s1, s2 = [int(x) for x in spec.split(":")]
    _lib.set_form("seg_lanes", lanes)
    ns = len(cfg["methods"])
    pit = [dict(tol=(2 * tol if s < ns - 1 else tol), **({"segments": (s1 if s == 0 else s2)} if (s1 if s == 0 else s2) else {})) for s in range(ns)]
    rx = bench.make_receiver(cfg, sig, tier="b", pit=pit)
    rx.load(sig)
    for _ in range(3):
        rx.run(overlap=True, prefetch=True)
    rx.wait_post(); _lib.sync()
    t0 = time.perf_counter()
    for _ in range(K):
        rx.run(overlap=True, prefetch=True)
    rx.wait_post(); _lib.sync()
    el = time.perf_counter() - t0
    reps = rx.pit_reports()
    tm = [rx.pit_timing[s][0] for s in range(ns)]
    print("%s tol %g lanes %d: %.1f MSym/s, %.3f ms per capture; S %s passes %s converged %s est %s pass_ms %s" %
          (key, tol, lanes, K * nsym / el / 1e6, el / K * 1e3, [r["segments"] for r in reps], [r["passes"] for r in reps], [r["converged"] for r in reps],
           [["%.2g" % d for d in r.get("deviation_rms", [])] for r in reps], [["%.3f" % x for x in t] for t in tm]), flush=True)
    del rx
_lib.set_form("seg_lanes", 0)
