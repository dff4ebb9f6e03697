"""Throughput of consecutive passes (tier b) with and without the phase search of pass k overlapped with the training of pass k+1.
Usage: python scripts/overlap_probe.py [workload ...]    (default: c3 ns c2)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from qampy_amd import _lib

for key in (sys.argv[1:] or ["c3", "ns", "c2"]):
    cfg = bench.WORKLOADS[key]
    nsym = cfg["nsym"]
    sig = bench.make_input(cfg, nsym, 1000, host=False)
    rx = bench.make_receiver(cfg, sig, tier="b", pit={})
    rx.load(sig)
    res = {}
    for ov in (False, True, False, True):
        for _ in range(2):
            rx.run(overlap=ov)
        rx.wait_post()
        _lib.sync()
        K = 12
        t0 = time.perf_counter()
        for _ in range(K):
            rx.run(overlap=ov)
        rx.wait_post()
        _lib.sync()
        el = (time.perf_counter() - t0) / K
        reps = rx.pit_reports()
        res.setdefault(ov, []).append(el * 1e3)
        print(key, "overlap" if ov else "serial ", "%.3f ms/step  %.1f MSym/s" % (el * 1e3, nsym / el / 1e6), "passes", [r["passes"] for r in reps],
              "certified", all(r["converged"] == 1 for r in reps), flush=True)
    rx.run(); a = rx.fetch()
    rx.run(overlap=True); rx.run(overlap=True); b = rx.fetch()
    print(key, "overlapped results identical to serial:", all(np.array_equal(a[k], b[k]) for k in ("out", "ph", "idx", "eq", "wxy")), flush=True)
