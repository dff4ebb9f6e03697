import sys, json, time
sys.path.insert(0, '.')
import numpy as np
import bench
from qampy_amd import _lib
_lib.init(0)
cfg = dict(bench.WORKLOADS["c3"])
nsym = 1 << 20
sig = bench.make_input(cfg, nsym, 1000)
ex = bench.make_receiver(cfg, sig); ex.load(sig); ex.run(); rex = ex.fetch()
print("exact ser", [e/n for e,n in bench.symbol_errors(rex["out"], sig)], "tapE", np.round(np.sum(np.abs(rex["wxy"])**2, axis=2), 3).tolist())
for pre, pmu, S in [((nsym, nsym), None, 1), ((1<<19, 1<<19), None, 2), ((1<<18,1<<18), None, 16), ((1<<16,1<<16), (1e-3,1e-3), 16), ((1<<16,1<<16), (1e-3,1e-3), 256)]:
    rx = bench.make_receiver(cfg, sig, segments=S, prefix=pre, prefix_mu=pmu); rx.load(sig)
    rx.run(); r = rx.fetch()
    e2 = r["err"][1]
    print(pre, pmu, S, "ser", [e/n for e,n in bench.symbol_errors(r["out"], sig)], "tapdiff %.3f" % np.max(np.abs(r["wxy"]-rex["wxy"])),
          "tapE", np.round(np.sum(np.abs(r["wxy"])**2, axis=2), 3).tolist(),
          "err2 pow by eighth", [[round(float(np.mean(np.abs(e2[m, i*e2.shape[1]//8:(i+1)*e2.shape[1]//8])**2)), 4) for i in range(8)] for m in range(2)], flush=True)
