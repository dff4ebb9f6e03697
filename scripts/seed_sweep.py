"""Tier b on the captures the ranks of an N-GPU run get (seed 1000 + rank, bench.py / sharding.channel_seed): every one must certify itself at the
headline tolerance WITHOUT the exact-form way out (0.3 s instead of 4 ms on that rank would set the whole job's time).  Usage: seed_sweep.py [workload] [tol] [nseeds]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from qampy_amd import _lib
key = sys.argv[1] if len(sys.argv) > 1 else "c3"
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
cfg = bench.WORKLOADS[key]
for seed in range(1000, 1000 + n):
    sig = bench.make_input(cfg, cfg["nsym"], seed)
    rx = bench.make_receiver(cfg, sig, tier="b", pit=dict(tol=tol))
    rx.load(sig)
    for _ in range(3):
        rx.run(overlap=True, prefetch=True)
    rx.wait_post(); _lib.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        rx.run(overlap=True, prefetch=True)
    rx.wait_post(); _lib.sync()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    reps = rx.pit_reports()
    errs = [d["errors"] for d in rx.ser(sig.symbols, maxlag=256, window=8192, trim=2000)]
    print("seed %d: %.3f ms per capture = %.0f MSym/s; passes %s exact_form %s est %s errors %s" % (
        seed, ms, cfg["nsym"] / ms / 1e3, [r["passes"] for r in reps], [r["exact_form"] for r in reps], ["%.2g" % r["deviation_rms"][-1] for r in reps], errs), flush=True)
    del rx, sig
