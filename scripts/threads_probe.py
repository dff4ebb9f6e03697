"""Several captures in flight on one GPU (pipeline.ReceiverGroup: one host thread and one stream set per receiver): aggregate throughput.
Usage: python scripts/threads_probe.py [workload] [passes per receiver] [receivers ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from qampy_amd import _lib
from qampy_amd.pipeline import ReceiverGroup, ResidentReceiver

key = sys.argv[1] if len(sys.argv) > 1 else "c3"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
counts = [int(a) for a in sys.argv[3:]] or [1, 2, 3]
cfg = bench.WORKLOADS[key]
nsym = cfg["nsym"]
sig = bench.make_input(cfg, nsym, 1000, host=False)
ref = None
for n in counts:
    g = ReceiverGroup(n, sig.shape[0], sig.shape[1], 2, cfg["M"], cfg["ntaps"], cfg["mu"], methods=cfg["methods"], Niter=cfg["niter"],
                      adaptive_stepsize=cfg["adaptive"], TrSyms=(None,) * len(cfg["methods"]), Mtestangles=cfg["A"], Nbps=cfg["Nbps"], dtype=np.complex64,
                      alphabet=sig.coded_symbols, tier="b", pit={})
    g.load(sig)
    g.run(3 * n)
    t0 = time.perf_counter()
    g.run(n * K)
    el = time.perf_counter() - t0
    outs = [rx.fetch() for rx in g.rx]
    if ref is None:
        ref = outs[0]
    same = all(np.array_equal(o[k], ref[k]) for o in outs for k in ("out", "ph", "idx", "eq", "wxy"))
    reps = g.pit_reports()
    print("%s: %d receiver(s) x %d captures in %.1f ms -> %.1f MSym/s, %.3f ms per capture; passes %s certified %s identical results %s" %
          (key, n, K, el * 1e3, n * K * nsym / el / 1e6, el / (n * K) * 1e3, [[r["passes"] for r in rp] for rp in reps],
           all(r["converged"] == 1 for rp in reps for r in rp), same), flush=True)
    g.close()
    del g
