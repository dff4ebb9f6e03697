"""Consecutive captures through one running receiver (overlap + prefetch), nothing else: for a rocprofv3 kernel trace of the steady state."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from qampy_amd import _lib
cfg = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4
sig = bench.make_input(cfg, cfg["nsym"], 1000)
rx = bench.make_receiver(cfg, sig, tier="b", pit=dict(tol=tol))
rx.load(sig)
for _ in range(n):
    rx.run(overlap=True, prefetch=os.environ.get("PREFETCH", "1") == "1")
rx.wait_post(); _lib.sync()
print([r["passes"] for r in rx.pit_reports()])
