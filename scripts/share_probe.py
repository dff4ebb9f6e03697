"""Several captures in flight, each receiver's segment grid sized for a SHARE of the chip (round 5): n receivers, one host thread and stream set each, the
passes of a receiver on 1024 / n of the SIMDs - what one capture's control path, prologue, acquisition and eigen-solver leave idle, the other captures'
passes use.  Usage: python scripts/share_probe.py [workload] [captures per receiver] [tol] [receivers ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from qampy_amd import _lib
from qampy_amd.pipeline import ReceiverGroup

key = sys.argv[1] if len(sys.argv) > 1 else "c3"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4
counts = [int(a) for a in sys.argv[4:]] or [1, 2, 3]
cfg = bench.WORKLOADS[key]
nsym = cfg["nsym"]
sig = bench.make_input(cfg, nsym, 1000, host=False)
for n in counts:
    for shared in ([False] if n == 1 else [False, True]):
        ns = len(cfg["methods"])
        if shared:
            pit = [dict(tol=(2 * tol if s < ns - 1 else tol), segments=int((896 if s == 0 else 1024) // n * 4 // 2)) for s in range(ns)]
        else:
            pit = dict(tol=tol)
        g = ReceiverGroup(n, sig.shape[0], sig.shape[1], 2, cfg["M"], cfg["ntaps"], cfg["mu"], methods=cfg["methods"], Niter=cfg["niter"],
                          adaptive_stepsize=cfg["adaptive"], TrSyms=(None,) * ns, Mtestangles=cfg["A"], Nbps=cfg["Nbps"], dtype=np.complex64,
                          alphabet=sig.coded_symbols, tier="b", pit=pit)
        g.load(sig)
        g.run(3 * n)
        t0 = time.perf_counter()
        g.run(n * K)
        el = time.perf_counter() - t0
        reps = g.pit_reports()
        print("%s tol %g: %d receiver(s)%s x %d captures -> %.1f MSym/s, %.3f ms per capture; S %s passes %s certified %s" %
              (key, tol, n, " each on 1/%d of the chip" % n if shared else "", K, n * K * nsym / el / 1e6, el / (n * K) * 1e3, [r["segments"] for r in reps[0]],
               [[r["passes"] for r in rp] for rp in reps], all(r["converged"] == 1 for rp in reps for r in rp)), flush=True)
        g.close()
        del g
