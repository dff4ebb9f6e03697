"""Per-pass estimates of tier b at the headline tolerance for given seeds, with the measured deviation from the exact path (C3)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from qampy_amd import _lib
cfg = bench.WORKLOADS["c3"]
tol = float(os.environ.get("TOL", 1e-4))
for seed in [int(a) for a in sys.argv[1:]] or [1001, 1007]:
    sig = bench.make_input(cfg, cfg["nsym"], seed)
    res = {}
    for tier in ("a", "b"):
        rx = bench.make_receiver(cfg, sig, tier=tier, pit=dict(tol=tol) if tier == "b" else None)
        rx.load(sig); rx.run(); res[tier] = rx.fetch(); res[tier + "rep"] = rx.pit_reports(); del rx
    a, b = res["a"], res["b"]
    print("seed", seed)
    for st in res["brep"]:
        print("   P %d tol %g | est rms %s | worst %s | taps %s | taps worst %s" % (st["passes"], st["tol"], ["%.3g" % v for v in st["deviation_rms"]], ["%.3g" % v for v in st["deviation"]],
              ["%.3g" % v for v in st["deviation_taps"]], ["%.3g" % v for v in st["deviation_taps_worst"]]))
    for m in range(2):
        print("   m%d eq %.2e taps %.2e err %s" % (m, np.sqrt(np.mean(np.abs(a["eq"][m] - b["eq"][m]) ** 2) / np.mean(np.abs(a["eq"][m]) ** 2)),
              np.linalg.norm(a["wxy"][m] - b["wxy"][m]) / np.linalg.norm(a["wxy"][m]), ["%.2e" % np.sqrt(np.mean(np.abs(a["err"][s][m] - b["err"][s][m]) ** 2)) for s in range(2)]))
