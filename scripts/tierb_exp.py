import sys, json, time
sys.path.insert(0, '.')
import numpy as np
import bench
from qampy_amd import _lib
_lib.init(0)
cfg = dict(bench.WORKLOADS["c3"])
sig = bench.make_input(cfg, cfg["nsym"], 1000)
ex = bench.make_receiver(cfg, sig); ex.load(sig); ex.run(); rex = ex.fetch()
print("exact ser", [e/n for e,n in bench.symbol_errors(rex["out"], sig)])
for pre in [(1<<16,1<<16), (1<<17,1<<16), (1<<18,1<<16), (1<<18,1<<17), (1<<18, 1<<18), (1<<19, 1<<18)]:
    for S in [256, 1024]:
        rx = bench.make_receiver(cfg, sig, segments=S, prefix=pre); rx.load(sig)
        rx.run(); _lib.sync(); t=time.perf_counter(); rx.run(); _lib.sync(); dt=time.perf_counter()-t
        r = rx.fetch()
        print(pre, S, "ms %.1f" % (dt*1e3), "ser", [e/n for e,n in bench.symbol_errors(r["out"], sig)], "tapdiff %.3f" % np.max(np.abs(r["wxy"]-rex["wxy"])), flush=True)
